#!/bin/bash
mkdir -p gpurun_out/r04; rm -f gpurun_out/r04/pool_c.err
timeout 300 python bench_pool.py --workers 4,8,16,32,64 --tiles 512 --broker 1 > gpurun_out/r04/pool_broker_lanes3.json 2>> gpurun_out/r04/pool_c.err
timeout 300 python bench_pool.py --workers 16,32,64 --tiles 512 --broker 1 --lanes 4 > gpurun_out/r04/pool_broker_lanes4.json 2>> gpurun_out/r04/pool_c.err
timeout 300 python bench_pool.py --workers 16,32,64 --tiles 512 --broker 1 --lanes 2 > gpurun_out/r04/pool_broker_lanes2.json 2>> gpurun_out/r04/pool_c.err
tail -5 gpurun_out/r04/pool_c.err
timeout 900 python -m pytest tests/test_gpu_pool.py tests/test_gpu_broker.py tests/test_gpu_fork.py -x -q > gpurun_out/r04/test_pool.txt 2>&1; tail -5 gpurun_out/r04/test_pool.txt
