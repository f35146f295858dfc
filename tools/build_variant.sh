#!/bin/bash
# Builds a kernel variant into raytrace_amd/lib/ab/<name>.so for tools/ab_bench.sh.
# usage: [SCENE_FLAGS="-DX=.."] tools/build_variant.sh name [-DFLAG=..]...   (flags go to rt_kernels.hip and rt_render.hip;
#        SCENE_FLAGS to rt_scene.cpp).  Result-changing measurement switches also need -DRT_MEASUREMENT_BUILD (rt_device.h).
set -e
name=$1; shift
cd "$(dirname "$0")/../raytrace_amd/csrc"
mkdir -p ../lib/ab /tmp/abbuild
HIPFLAGS="-std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc -O3 "$@" $HIPFLAGS -c -o /tmp/abbuild/$name.o rt_kernels.hip
/opt/rocm/bin/hipcc -O3 "$@" $HIPFLAGS -c -o /tmp/abbuild/${name}_render.o rt_render.hip
scene=rt_scene.o
if [ -n "$SCENE_FLAGS" ]; then
  g++ $SCENE_FLAGS -O2 -std=c++17 -fPIC -pthread -ffp-contract=off -Wall -c -o /tmp/abbuild/${name}_scene.o rt_scene.cpp
  scene=/tmp/abbuild/${name}_scene.o
else
  [ -f rt_scene.o ] || make rt_scene.o
fi
[ -f rt_abi.o ] || make rt_abi.o
/opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o ../lib/ab/$name.so $scene rt_abi.o /tmp/abbuild/$name.o /tmp/abbuild/${name}_render.o
echo built $name
