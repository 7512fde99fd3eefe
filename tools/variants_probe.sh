#!/bin/bash
# tools/variants_probe.sh -- GPU box: time the probe builds of tools/build_variants.sh (S2P_HIP_LIB) against the shipped library
cd "$(dirname "$0")/.."
one() {  # name lib args...
  local name=$1 lib=$2; shift 2
  echo "[$name] $*: $(S2P_HIP_LIB=$lib python bench.py --no-cpu --no-job --no-pool --steps 3 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile | aggregate launch %.4f ms' % (d['ms_per_tile'], d['stage_ms']['aggregate']))")"
}
LONE="--batch 24 --batch-launch 1 --streams 1"
echo "== the lone launch (VERDICT r03 item 4): the axis lattices alone, 8-wave bands vs 4-wave / rings of 16 / successor published at step 8"
for v in axis8 axis8t8 axis8r16 axis4 axis4r16t8; do one $v build/variants/libs2p_hip_$v.so $LONE; done
one shipped "" $LONE
echo "== 4-wave bands for the batches that still run 8 (item 5)"
for v in shipped nw4; do
  lib=""; [ $v != shipped ] && lib=build/variants/libs2p_hip_$v.so
  one $v "$lib" --workload config3 --batch-launch 1 --streams 1
  one $v "$lib" --workload config3
  one $v "$lib" --workload config3 --batch-launch 4
  one $v "$lib" --size 1024 --ndisp 512 --batch 24 --batch-launch 1 --streams 1
  one $v "$lib" --size 1024 --ndisp 512 --batch 24 --batch-launch 4 --streams 2
  one $v "$lib" --size 512 --ndisp 256 --batch 64 --batch-launch 8 --streams 2
  echo "[$v] config4 job: $(S2P_HIP_LIB=$lib python bench.py --workload config4 --no-cpu --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile' % d['ms_per_step'])")"
done
