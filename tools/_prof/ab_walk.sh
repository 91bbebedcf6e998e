# builds A/B variants of the library for the chunk-walking Schur kernel (switches in schur_walk.hip.h) into tools/_prof/ab/ — run here, then
# on the GPU box:  for v in ...; do AB_LIB=tools/_prof/ab/libmpcg_$v.so python tools/time_schur.py 128 1024 | grep "ss .*L=16"; done
mkdir -p tools/_prof/ab
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function $2 mpcgpu_amd/csrc/mpcg_*.hip -o tools/_prof/ab/libmpcg_$1.so & }
for v in "$@"; do
  case $v in
    base) build base "-DSW_PREFETCH=0 -DSW_PAIR=0" ;;
    nostore) build nostore "-DSW_PREFETCH=0 -DSW_PAIR=0 -DSW_ABLATE=1" ;;
    noload) build noload "-DSW_PREFETCH=0 -DSW_PAIR=0 -DSW_ABLATE=2" ;;
    noio) build noio "-DSW_PREFETCH=0 -DSW_PAIR=0 -DSW_ABLATE=3" ;;
    noio_nogj) build noio_nogj "-DSW_PREFETCH=0 -DSW_PAIR=0 -DSW_ABLATE=7" ;;
    noio_nogemm) build noio_nogemm "-DSW_PREFETCH=0 -DSW_PAIR=0 -DSW_ABLATE=11" ;;
    noio_none) build noio_none "-DSW_PREFETCH=0 -DSW_PAIR=0 -DSW_ABLATE=15" ;;
    *) build $v "$AB_FLAGS" ;;
  esac
done
wait
ls -la tools/_prof/ab/
