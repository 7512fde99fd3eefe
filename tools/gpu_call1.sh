set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles/r05
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/profiles/r05/gpu_tests_call1.txt
tail -5 gpurun_out/profiles/r05/gpu_tests_call1.txt
timeout 300 tools/pmc_calib.sh 2>&1 | tail -60
timeout 600 tools/noc_valu_probe.sh 2>&1 | tail -12
