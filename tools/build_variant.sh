#!/bin/bash
# build a variant of libtracer_amd.so with extra compiler flags: tools/build_variant.sh <name> <flags...>
# -> tracer_amd/variants/libtracer_<name>.so   (select with TRC_LIB=... at run time)
set -e
cd "$(dirname "$0")/../tracer_amd/csrc"
name=$1; shift
mkdir -p ../variants /tmp/trcvar_$name
F="--offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -Wno-unused-function -Wno-unknown-pragmas $*"
/opt/rocm/bin/hipcc $F -c kernels.hip -o /tmp/trcvar_$name/kernels.o &
/opt/rocm/bin/hipcc $F -c tracer_abi.hip -o /tmp/trcvar_$name/tracer_abi.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../variants/libtracer_$name.so /tmp/trcvar_$name/kernels.o /tmp/trcvar_$name/tracer_abi.o host/SceneApi.o host/Camera.o host/Renderer.o host/scenes.o
echo built ../variants/libtracer_$name.so
