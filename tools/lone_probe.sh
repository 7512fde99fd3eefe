#!/bin/bash
# tools/lone_probe.sh -- GPU box: the lone MGM launch.  The launch IS the chain of its 4 axis lattices (0.95 of
# 0.97 ms), so the axis lattices alone (-DS2P_MGM_ONLY_AXIS) are the upper bound of what giving THEM 16-row bands (4 waves), rings of
# 16 entries and an earlier publication of the successor (step 8) could buy, with the diagonal lattices left as they are.
cd "$(dirname "$0")/.."
one() {
  local name=$1 lib=$2; shift 2
  echo "[$name] $(S2P_HIP_LIB=$lib python bench.py --no-cpu --no-job --no-pool --steps 3 --batch 24 --batch-launch 1 --streams 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lone launch %.4f ms (tile %.4f ms)' % (d['stage_ms']['aggregate'], d['ms_per_tile']))")"
}
for v in axis8 axis8t8 axis8r16 axis4 axis4r16 axis4r16t8; do one $v build/variants/libs2p_hip_$v.so; done
one shipped ""
