#!/bin/bash
# tools/sq_probe.sh [ROUND] -- SQ counters of the shipped kernels of the headline's call shape (8 tiles per library call, 1024^2 x 128, MGM
# recursion with three predecessors, confidence image) and of the lone launch, averaged per launch -> profiles/<ROUND>/sq_counters_mgm.txt.
# Counters in their own passes (--pmc with --kernel-trace only, a group of <= 8 SQ counters per pass), as MI355X_MICROARCH.md prescribes.
# Units: SQ_*_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* are quad-cycles summed over waves (per SIMD); SQ_BUSY_CYCLES is per shader engine
# (32 of them): / 32 = cycles of the launch.
cd "$(dirname "$0")/.."
ROUND=${1:-r06}
OUT=gpurun_out/profiles/$ROUND
mkdir -p $OUT
export TMPDIR=/tmp
GROUPS_=(
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH"
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"
  "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_SENDMSG SQ_INST_LEVEL_VMEM SQ_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_I8"
)
{
echo "# tools/sq_probe.sh $ROUND: SQ counters on 1024x1024x128, MGM recursion (three predecessors), confidence image; average per launch of each kernel."
echo "# b8 = the headline's call shape (8 tiles per library call, one call at a time); b1 = one tile per call.  Library: $(python -c 'from s2p_amd import _lib; print(_lib.lib().s2p_hip_build_info().decode())')"
echo "# label kernel counter launches average"
for wl in "b8|--recursion 2 --streams 1 --batch-launch 8 --batch 16" "b1|--recursion 2 --streams 1 --batch-launch 1 --batch 6"; do
  label=${wl%%|*}; args=${wl#*|}
  CMD="python bench.py $args --steps 2 --warmup 1 --no-cpu --no-job --no-pool"
  g=0
  for grp in "${GROUPS_[@]}"; do
    g=$((g + 1))
    rm -rf gpurun_out/sq_${label}_$g
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/sq_${label}_$g -- $CMD > /dev/null 2>&1
    python - "$label" "gpurun_out/sq_${label}_$g" <<'EOP'
import csv, glob, sys, collections
label, d = sys.argv[1], sys.argv[2]
fs = glob.glob(d + "/*/*counter_collection.csv")
if not fs:
    print("# %s: no counter file in %s (group refused?)" % (label, d)); sys.exit(0)
per = collections.defaultdict(float)          # (kernel, counter, dispatch) -> sum over the rows of a dispatch (one per instance, if the tool splits them)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    short = "k_mgm_bands" if "k_mgm_bands" in k else "k_wta_census_pk" if "k_wta_census_pk" in k else "k_census_cost" if "k_census_cost" in k else "k_median_valid" if "k_median_valid" in k else None
    if short:
        per[(short, r["Counter_Name"], r.get("Dispatch_Id", r.get("Correlation_Id", "")))] += float(r["Counter_Value"])
acc = collections.defaultdict(list)
for (k, c, _), v in per.items():
    acc[(k, c)].append(v)
for (k, c), v in sorted(acc.items()):
    print("%s %s %s %d %.1f" % (label, k, c, len(v), sum(v) / len(v)))
EOP
    rm -rf gpurun_out/sq_${label}_$g
  done
done
} 2>&1 | tee $OUT/sq_counters_mgm.txt
