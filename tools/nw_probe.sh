#!/bin/bash
# tools/nw_probe.sh -- GPU box: 4-wave bands (-DS2P_MGM_NW_WIDE=4) against the shipped 8-wave ones for the layouts whose batches
# still run 8 waves (D = 256 with 16 disparities per lane, D = 512); the D = 128 batch already runs 4 (profiles/r03/nw4_probe.txt).
# Parity subset under the flag first: the flag changes every G >= 16 layout.
cd "$(dirname "$0")/.."
run() {
  for ARGS in "--workload config3 --batch-launch 1 --streams 1" "--workload config3" "--size 1024 --ndisp 512 --batch 24 --batch-launch 1 --streams 1" "--size 1024 --ndisp 512 --batch 24 --batch-launch 4 --streams 2"; do
    echo "[$1] $ARGS: $(python bench.py --no-cpu --no-job --steps 3 $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile | aggregate launch %.4f ms' % (d['ms_per_tile'], d['stage_ms']['aggregate']))")"
  done
  echo "[$1] config4 job: $(python bench.py --workload config4 --no-cpu --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile' % d['ms_per_step'])")"
}
run "shipped"
S2P_HIP_EXTRA_FLAGS="-DS2P_MGM_NW_WIDE=4" python -m s2p_amd.build --force > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_mgm_bands.py tests/test_gpu_batch.py tests/test_gpu_jobs.py -x -q -m gpu 2>&1 | tail -2
run "-DS2P_MGM_NW_WIDE=4"
python -m s2p_amd.build --force > /dev/null 2>&1
