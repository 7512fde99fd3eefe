#!/bin/bash
# round 4: 16 directions -- the new parity cases first, then the stage times
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_census.py tests/test_gpu_batch.py tests/test_gpu_warp_and_tile.py tests/test_gpu_mgm_bands.py -m gpu -q -k "nb_dir or 27 or 28 or error_statuses or scheduler_and_file or bands_match or full_size or nb_dir4 or different_shapes_in_one or multi_scale_batch_equals or batch_equals" > gpurun_out/r04/gpu_dir16.txt 2>&1; tail -15 gpurun_out/r04/gpu_dir16.txt
timeout 600 python tools/dir16_time.py > gpurun_out/r04/dir16_time.txt 2>&1; cat gpurun_out/r04/dir16_time.txt
