#!/bin/bash
# tools/inner_probe.sh -- GPU box: the MGM launch with / without the unmasked inner blocks (S2P_MGM_INNER) and the prologue's
# paired stores (S2P_MGM_PROLOGUE_STORES: what the compiler's vmcnt waits can prove), parity first
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_mgm_bands.py tests/test_gpu_census.py tests/test_gpu_batch.py tests/test_gpu_jobs.py -x -q -m gpu 2>&1 | tail -3
run() {
  for ARGS in "--batch 96 --batch-launch 1 --streams 1" "--batch 96 --batch-launch 8 --streams 1" "--batch 96" "--workload config3 --batch-launch 1 --streams 3" "--workload config3" "--size 1024 --ndisp 512 --batch 24 --batch-launch 1 --streams 1"; do
    echo "[$1] $ARGS: $(python bench.py --no-cpu --no-job --steps 3 $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms per tile | aggregate launch %.4f ms' % (d['ms_per_tile'], d['stage_ms']['aggregate']))")"
  done
}
run "shipped"
for F in "$@"; do
  S2P_HIP_EXTRA_FLAGS="$F" python -m s2p_amd.build --force > /dev/null 2>&1
  run "$F"
done
python -m s2p_amd.build --force > /dev/null 2>&1
