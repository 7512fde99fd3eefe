#!/bin/bash
# upper bound of "the consensus comes out of the band kernel": WTA without its per-direction arg-mins (confidence image NOT produced: timing
# only) and / or 10 more VALU instructions per step of the band kernel, in the regime that counts (three streams, 8 tiles per call)
cd "$(dirname "$0")/.."
for rep in 1 2; do for V in shipped fakeconf fakeconf_v10 v10; do
  if [ $V = shipped ]; then unset S2P_HIP_LIB; else export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so; fi
  echo "$V: $(python bench.py --steps 10 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f G/s, band launch %.3f ms' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3, d['roofline']['avg_launch_ms']))")"
done; done
