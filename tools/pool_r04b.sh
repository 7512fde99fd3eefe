#!/bin/bash
mkdir -p gpurun_out/r04
timeout 300 python -m pytest tests/test_gpu_broker.py -x -q > gpurun_out/r04/test_broker.txt 2>&1; tail -5 gpurun_out/r04/test_broker.txt
timeout 600 python bench_pool.py --workers 1,4,8,16,32,64 --tiles 256 --broker 1 > gpurun_out/r04/pool_broker_sweep.json 2> gpurun_out/r04/pool_broker_sweep.err
tail -3 gpurun_out/r04/pool_broker_sweep.err
cat /tmp/s2p_hip_broker_0/gpu0.log 2>/dev/null | tail -20
timeout 900 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r04/test_pool.txt 2>&1; tail -15 gpurun_out/r04/test_pool.txt
