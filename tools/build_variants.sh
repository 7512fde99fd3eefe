#!/bin/bash
# tools/build_variants.sh NAME "FLAGS" [NAME "FLAGS" ...] -- PROBE builds of libs2p_hip.so side by side with the shipped one:
# build/variants/libs2p_hip_NAME.so (selected at run time with S2P_HIP_LIB=...; they travel to the GPU box with the snapshot, so one
# gpurun call can time several builds without compiling there).  Goes through s2p_amd/build.py, which marks every such library as a
# probe build (s2p_hip_build_info(), the "[PROBE BUILD]" banner of its error messages) and never writes one to s2p_amd/lib/.
cd "$(dirname "$0")/.."
python -m s2p_amd.build > /dev/null          # the shipped library
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  ( S2P_HIP_VARIANT=$NAME S2P_HIP_EXTRA_FLAGS="$FLAGS" python -m s2p_amd.build > /dev/null && echo "built $NAME ($FLAGS)" ) &
done
wait
ls -la build/variants/*.so
