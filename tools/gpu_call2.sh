set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles/r05
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/profiles/r05/gpu_tests_call2.txt
tail -8 gpurun_out/profiles/r05/gpu_tests_call2.txt
timeout 700 tools/direct_hang_probe.sh 3 2>&1 | tail -40
timeout 900 tools/collect_r05.sh 2>&1 | tail -30
cat gpurun_out/profiles/r05/bench_default_1gpu.json | head -c 3000
