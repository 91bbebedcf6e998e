# Builds of the library with pieces of generate_kkt_kernel compiled out (timing experiments; results are wrong).  Here:  bash tools/kkt_ablate.sh
# then on the GPU box:  for m in 0 1 2 4 8 16 17; do AB_LIB=tools/_prof/ab/libmpcg_kkt_abl$m.so python tools/kkt_time.py; done
mkdir -p tools/_prof/ab
for m in 0 1 2 4 8 16 17; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function -DKKT_ABLATE=$m mpcgpu_amd/csrc/mpcg_*.hip -o tools/_prof/ab/libmpcg_kkt_abl$m.so 2>/dev/null &
done
wait
ls -la tools/_prof/ab/ | grep kkt_abl
