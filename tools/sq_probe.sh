#!/bin/bash
# tools/sq_probe.sh -- GPU box: SQ counters of k_mgm_bands, one tile per launch and 8 per launch (what is the full chip waiting for?)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/sq_probe; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/avail.txt
WANT="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_CYCLES SQ_INSTS_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_WAVE32_LDS SQ_WAVE_READY SQ_WAVE_DEP_STALL SQ_WAIT_IFETCH SQ_ACCUM_PREV"
for wl in "b1|--batch-launch 1 --batch 6" "b8|--batch-launch 8 --batch 16"; do
  name=${wl%%|*}; args=${wl#*|}
  for c in $WANT; do
    grep -qx $c $OUT/avail.txt || continue
    rm -rf gpurun_out/pmc_sq
    timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_sq -- python bench.py --recursion 2 --streams 1 $args --steps 2 --warmup 1 --no-cpu --no-job > /dev/null 2>&1
    python - $name $c <<'EOP' >> $OUT/sq_$name.txt
import csv, glob, sys
name, c = sys.argv[1:3]
f = glob.glob("gpurun_out/pmc_sq/*/*counter_collection.csv")
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c and "k_mgm_bands" in r["Kernel_Name"]] if f else []
print(name, c, len(v), sum(v) / max(1, len(v)))
EOP
  done
done
rm -rf gpurun_out/pmc_sq
cat $OUT/sq_b1.txt $OUT/sq_b8.txt
