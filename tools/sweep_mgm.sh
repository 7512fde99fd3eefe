#!/bin/bash
# tools/sweep_mgm.sh build|run|trace  -- A/B of the band-pipelined MGM kernel's compile-time knobs.
#   build: (here, no GPU) compile one libs2p_hip.so per variant into build/variants/<name>/  (build/ travels with gpurun)
#   run:   (GPU box) bench every variant in MGM mode (1 and 2 tile streams), one line each -> gpurun_out/sweep_mgm.txt
#   trace: (GPU box) per-band time stamps of the variants whose name ends in _trace -> gpurun_out/trace_<name>.txt
# Variants: "<name> <extra hipcc flags>" below; build/variants/r01 (if present) is the previous round's library.
set -e
cd "$(dirname "$0")/.."
SRC="s2p_amd/csrc/api.hip s2p_amd/csrc/sgbm_kernels.hip s2p_amd/csrc/census_kernels.hip s2p_amd/csrc/warp_kernels.hip s2p_amd/csrc/tri_kernels.hip s2p_amd/csrc/fusion_kernels.hip s2p_amd/csrc/raster_kernels.hip"
VARIANTS=(
  "cur"
  "cur_trace -DS2P_MGM_TRACE"
  "q0 -DS2P_MGM_ONLY_Q0"
  "q0_trace -DS2P_MGM_TRACE -DS2P_MGM_ONLY_Q0"
)
case "$1" in
build)
  for v in "${VARIANTS[@]}"; do
    set -- $v; name=$1; shift
    mkdir -p build/variants/$name
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -fvisibility=hidden "$@" \
      -o build/variants/$name/libs2p_hip.so $SRC &
  done
  wait
  ls -la build/variants/*/libs2p_hip.so
  ;;
run)
  mkdir -p gpurun_out
  cp s2p_amd/lib/libs2p_hip.so build/libs2p_hip.orig.so
  : > gpurun_out/sweep_mgm.txt
  for d in build/variants/*/; do
    name=$(basename $d)
    case $name in *_trace) continue;; esac
    cp $d/libs2p_hip.so s2p_amd/lib/libs2p_hip.so
    for st in ${STREAMS:-1 2}; do
      timeout 120 python bench.py --algo census --recursion 1 --streams $st --steps ${STEPS:-30} --warmup 4 --no-cpu ${BENCH_ARGS:-} 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name streams=$st ms/tile', d['ms_per_step'], 'agg', d['stage_ms']['aggregate'], 'wta', d['stage_ms']['wta'])" \
        | tee -a gpurun_out/sweep_mgm.txt || echo "$name streams=$st FAILED" | tee -a gpurun_out/sweep_mgm.txt
    done
  done
  cp build/libs2p_hip.orig.so s2p_amd/lib/libs2p_hip.so
  ;;
trace)
  mkdir -p gpurun_out
  cp s2p_amd/lib/libs2p_hip.so build/libs2p_hip.orig.so
  for d in build/variants/*_trace/; do
    name=$(basename $d)
    cp $d/libs2p_hip.so s2p_amd/lib/libs2p_hip.so
    timeout 120 python tools/mgm_trace.py run 2> gpurun_out/trace_raw_$name.log || true
    python tools/mgm_trace.py < gpurun_out/trace_raw_$name.log > gpurun_out/trace_$name.txt || true
    rm -f gpurun_out/trace_raw_$name.log
  done
  cp build/libs2p_hip.orig.so s2p_amd/lib/libs2p_hip.so
  ;;
*) echo "usage: $0 build|run|trace"; exit 2;;
esac
