#!/bin/bash
# tools/build_variants.sh NAME "FLAGS" [NAME "FLAGS" ...] -- probe builds of libs2p_hip.so side by side with the shipped one:
# build/variants/libs2p_hip_NAME.so (selected at run time with S2P_HIP_LIB=...; they travel to the GPU box with the snapshot,
# so one gpurun call can time several builds without compiling there).  Only census_kernels.hip depends on the MGM flags.
cd "$(dirname "$0")/.."
mkdir -p build/variants
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CF="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -fvisibility=hidden"
python -m s2p_amd.build > /dev/null          # the shipped objects
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  ( mkdir -p build/variants/obj_$NAME
    $HIPCC $CF $FLAGS -c s2p_amd/csrc/census_kernels.hip -o build/variants/obj_$NAME/census_kernels.o &&
    $HIPCC --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o build/variants/libs2p_hip_$NAME.so \
        build/obj/api.o build/obj/sgbm_kernels.o build/variants/obj_$NAME/census_kernels.o build/obj/warp_kernels.o build/obj/tri_kernels.o \
        build/obj/fusion_kernels.o build/obj/raster_kernels.o && echo "built $NAME ($FLAGS)" ) &
done
wait
ls -la build/variants/*.so
