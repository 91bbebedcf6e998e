#!/bin/bash
# A variant build of the library for A/B runs: tools/_prof/build_variant.sh <name> <extra flags for mpcg_pcg.hip...>
#   -> tools/_prof/libmpcg_hip_<name>.so  (the other translation units are the tree's objects; run `make lib` first)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c mpcgpu_amd/csrc/mpcg_pcg.hip -o /tmp/mpcg_pcg_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/mpcg_pcg_$name.o mpcgpu_amd/csrc/mpcg_producers.o mpcgpu_amd/csrc/mpcg_plant.o mpcgpu_amd/csrc/mpcg_ldl.o -o tools/_prof/libmpcg_hip_$name.so
echo built tools/_prof/libmpcg_hip_$name.so
