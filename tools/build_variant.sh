#!/bin/bash
# Builds a kernel variant into raytrace_amd/lib/ab/<name>.so for tools/ab_bench.sh.  usage: tools/build_variant.sh name [-DFLAG=..]...
set -e
name=$1; shift
cd "$(dirname "$0")/../raytrace_amd/csrc"
mkdir -p ../lib/ab /tmp/abbuild
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -c -o /tmp/abbuild/$name.o rt_kernels.hip
[ -f rt_scene.o ] || make rt_scene.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/ab/$name.so rt_scene.o /tmp/abbuild/$name.o
echo built $name
