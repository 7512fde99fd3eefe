#!/bin/bash
# round 4, first measurement of the Pool model: cold start anatomy, then P sweep, then the stress test
mkdir -p gpurun_out/r04
df -h /dev/shm | tail -1 > gpurun_out/r04/shm.txt; nproc >> gpurun_out/r04/shm.txt; free -g | head -2 >> gpurun_out/r04/shm.txt
python tools/cold_start.py > gpurun_out/r04/cold_start_1proc.txt 2>&1
python tools/cold_start.py --procs 16 > gpurun_out/r04/cold_start_16procs.txt 2>&1
python tools/cold_start.py --procs 64 > gpurun_out/r04/cold_start_64procs.txt 2>&1
timeout 600 python bench_pool.py --workers 1,2,4,8,16,32,64 --tiles 256 > gpurun_out/r04/pool_sweep.json 2> gpurun_out/r04/pool_sweep.err
timeout 600 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r04/test_pool.txt 2>&1
tail -3 gpurun_out/r04/test_pool.txt
cat gpurun_out/r04/cold_start_1proc.txt
