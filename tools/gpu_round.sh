#!/bin/bash
# tools/gpu_round.sh -- one gpurun call that produces a round's evidence on the tree it is run on:
#   gpurun --timeout 2100 -- 'bash tools/gpu_round.sh'
# GPU tests (tail kept), the direct-mode Pool with the process fence lifted (three runs), then tools/collect_r05.sh: kernel-trace
# stats + separate PMC passes of the headline / lone / 8-path call shapes, the in-flight trace of the headline command, the default
# bench line and the job lines.  Everything lands under gpurun_out/profiles/r05/ (copy what is judged into profiles/r05/).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles/r05
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/profiles/r05/gpu_tests_tail.txt
tail -8 gpurun_out/profiles/r05/gpu_tests_tail.txt
timeout 500 tools/direct_hang_probe.sh 3 2>&1 | tail -40
timeout 900 tools/collect_r05.sh 2>&1 | tail -30
head -c 3000 gpurun_out/profiles/r05/bench_default_1gpu.json
