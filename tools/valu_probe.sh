#!/bin/bash
# (record of one measurement: the S2P_WTA_PAIRS switch its "nopairs" / "base" builds used was removed with the code after this run)
# round 4: VALU savings measured where they count -- the bench headline (tiles in flight on three streams).  Builds: shipped (pair-packed
# consensus keys + kill mask in the cost kernel), nopairs (-DS2P_WTA_PAIRS=0), base (both off): tools/build_variants.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04
timeout 400 python -m pytest tests/test_gpu_census.py tests/test_gpu_batch.py -m gpu -q -x > gpurun_out/r04/gpu_valu.txt 2>&1; tail -2 gpurun_out/r04/gpu_valu.txt
for rep in 1 2; do for V in shipped nopairs base; do
  if [ $V = shipped ]; then unset S2P_HIP_LIB; else export S2P_HIP_LIB=$PWD/build/variants/libs2p_hip_$V.so; fi
  echo "$V: $(python bench.py --steps 10 --warmup 3 --no-job --no-pool --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile, %.1f G/s' % (d['ms_per_step'] / d['config']['tiles_per_step'], d['value'] / 1e3))")"
done; done
unset S2P_HIP_LIB
python tools/conf_time.py
