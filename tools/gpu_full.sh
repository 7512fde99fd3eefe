#!/bin/bash
# the round-end sequence: GPU tests, smoke, default bench line
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_tests.txt 2>&1; tail -6 gpurun_out/r04/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.txt 2>&1; tail -2 gpurun_out/r04/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_default_1gpu.json 2> gpurun_out/r04/bench_default.err; tail -c 600 gpurun_out/r04/bench_default.err
ls /tmp/s2p_hip_broker_0/ 2>/dev/null; pgrep -af "s2p_amd.broker" || true
