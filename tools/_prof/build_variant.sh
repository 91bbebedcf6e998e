#!/bin/bash
# A variant build of the library for A/B runs: tools/_prof/build_variant.sh <name> [--tu mpcg_plant] <extra flags for that translation unit...>
#   -> tools/_prof/libmpcg_hip_<name>.so  (the other translation units are the tree's objects; run `make lib` first).  Default unit: mpcg_pcg.
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
tu=mpcg_pcg
if [ "$1" = "--tu" ]; then tu=$2; shift 2; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c mpcgpu_amd/csrc/$tu.hip -o /tmp/${tu}_$name.o
objs=""
for u in mpcg_pcg mpcg_producers mpcg_plant mpcg_ldl; do
  if [ $u = $tu ]; then objs="$objs /tmp/${tu}_$name.o"; else objs="$objs mpcgpu_amd/csrc/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/_prof/libmpcg_hip_$name.so
echo built tools/_prof/libmpcg_hip_$name.so
