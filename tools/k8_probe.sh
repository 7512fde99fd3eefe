#!/bin/bash
# tools/k8_probe.sh -- GPU box: 16 disparities per lane (K = 8) in the MGM band kernel from D = 256 / 512 / never (-DS2P_MGM_K8_FROM):
# rebuilds the library per setting and times the configs[3] shape alone, in flight and batched (profiles/r03/k8_probe.txt)
for FROM in 4096 256 512; do
  S2P_HIP_EXTRA_FLAGS="-DS2P_MGM_K8_FROM=$FROM" python -m s2p_amd.build --force > /dev/null 2>&1
  for ARGS in "--workload config3 --batch-launch 1 --streams 1" "--workload config3 --batch-launch 1 --streams 3" "--workload config3" "--size 1024 --ndisp 512 --batch 24 --batch-launch 1 --streams 1" "--size 1024 --ndisp 512 --batch 24 --batch-launch 4 --streams 2" "--size 512 --ndisp 256 --batch-launch 1 --streams 1" "--size 512 --ndisp 256"; do
    echo "K8 from $FROM | $ARGS: $(python bench.py --no-cpu --no-job --steps 2 $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"ms_per_tile\"], d[\"stage_ms\"][\"aggregate\"])")"
  done
done
S2P_HIP_EXTRA_FLAGS="-DS2P_MGM_K8_FROM=256" python -m s2p_amd.build --force > /dev/null 2>&1
python -m pytest tests/test_gpu_mgm_bands.py tests/test_gpu_census.py tests/test_gpu_batch.py tests/test_gpu_jobs.py -x -q -m gpu 2>&1 | tail -3
python -m s2p_amd.build --force > /dev/null 2>&1
