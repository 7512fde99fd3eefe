#!/bin/bash
# tools/gpu_round.sh [ROUND] -- one gpurun call that produces a round's evidence on the tree it is run on:
#   gpurun --timeout 2700 -- 'bash tools/gpu_round.sh r06'
# GPU tests (tail kept), then tools/collect_round.sh: calibration, kernel-trace stats + separate PMC passes of every call shape the bench
# line reports, the in-flight trace of the headline command, SQ counters, the default bench line and the job lines.  Everything lands
# under gpurun_out/profiles/<ROUND>/ (copy what is judged into profiles/<ROUND>/).
set -x
cd $GRAFT_REPO_ROOT
ROUND=${1:-r06}
mkdir -p gpurun_out/profiles/$ROUND
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/profiles/$ROUND/gpu_tests_tail.txt
tail -8 gpurun_out/profiles/$ROUND/gpu_tests_tail.txt
timeout 1400 tools/collect_round.sh $ROUND 2>&1 | tail -30
head -c 3000 gpurun_out/profiles/$ROUND/bench_default_1gpu.json
