# build ablated variants of the library (here, on the CPU box) — then on the GPU box:  for m in ...; do AB_LIB=... python tools/lpb_ablate.py; done
mkdir -p tools/_prof/ab
for m in 0 1 2 3 4 7 8 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function -DMPCG_ABLATE=$m mpcgpu_amd/csrc/mpcg_capi.hip -o tools/_prof/ab/libmpcg_abl$m.so &
done
wait
ls -la tools/_prof/ab/
