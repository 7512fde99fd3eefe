#!/bin/bash
# tools/e34_probe.sh -- GPU box: the UPPER BOUND of what packing e to 6 bits / C to 5 bits (26 -> 19.75 B per candidate)
# could buy: a build that moves those bytes WITHOUT any packing instruction (every 4th e-store of the aggregation dropped, 6 of the 8
# e-volumes read by the WTA; results invalid) against the shipped library, on the lone launch, the 8-tile launch and whole tiles.
cd "$(dirname "$0")/.."
one() {
  local name=$1 lib=$2; shift 2
  echo "[$name] $*: $(S2P_HIP_LIB=$lib python bench.py --no-cpu --no-job --no-pool --steps 4 "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.4f ms per tile | cost %.4f aggregate launch %.4f wta %.4f (per call of %d)' % (d['ms_per_tile'], s['cost'], s['aggregate'], s['wta'], d['config']['tiles_per_call']))")"
}
for v in shipped e34 shipped e34; do
  lib=""; [ $v != shipped ] && lib=build/variants/libs2p_hip_$v.so
  one $v "$lib" --batch 24 --batch-launch 1 --streams 1
  one $v "$lib" --batch 128 --batch-launch 8 --streams 1
  one $v "$lib" --batch 256
done
