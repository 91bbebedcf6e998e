# Builds of the library with pieces of pcg_lpk_kernel compiled out (timing experiments; results are wrong): bit 1 = no off-diagonal work
# (transposed + direct-L products, z stores), bit 2 = no diagonal columns 1..6, bit 4 = no wave fold.  Here:  bash tools/lpk_ablate.sh
# then on the GPU box:  for m in 0 1 3 7; do AB_LIB=tools/_prof/ab/libmpcg_lpk_abl$m.so python tools/lpk_quick.py 128x1024 128x1; done
mkdir -p tools/_prof/ab
for m in 0 1 3 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function -DMPCG_ABLATE_LPK=$m mpcgpu_amd/csrc/mpcg_*.hip -o tools/_prof/ab/libmpcg_lpk_abl$m.so &
done
wait
ls -la tools/_prof/ab/ | grep lpk
