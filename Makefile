# Top-level build of the drop-in boundary — the counterpart of the reference's Makefile
# (/root/reference/Makefile:1-20: `make examples` builds examples/pcg.exe and examples/qdldl.exe with nvcc).
#
#   make            libmpcg_hip.so (HIP kernels + C ABI, gfx950) and the C++ call-site programs over the shim headers
#   make lib        the library only
#   make examples   the call-site programs (LINSYS_SOLVE = 1 and 0, float and -DUSE_DOUBLES)
#   make oracle     the CPU checker (gcc; test infrastructure, never linked into the product)
#   make clean
#
# hipcc cross-compiles gfx950 without a GPU.  mpcgpu_amd/build.py (python -m mpcgpu_amd.build, __graft_entry__.build()) runs this file.

HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG     := mpcgpu_amd
CSRC    := $(PKG)/csrc
LIB     := $(PKG)/libmpcg_hip.so
INC     := include

LIBFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
EXFLAGS  := --offload-arch=$(ARCH) -O2 -std=c++17
RPATH    := -Wl,-rpath,'$$ORIGIN/../$(PKG)'
LINKLIB  := -L$(PKG) -lmpcg_hip $(RPATH)

LIB_SRCS := $(wildcard $(CSRC)/*.hip)
LIB_OBJS := $(patsubst $(CSRC)/%.hip,$(CSRC)/%.o,$(LIB_SRCS))
LIB_DEPS := $(wildcard $(CSRC)/*.h $(CSRC)/*.hpp $(CSRC)/*.inc) $(INC)/mpcg.h

SHIM_PCG   := $(INC)/gbd_pcg_compat/gpu_pcg.cuh $(INC)/gbd_pcg_compat/gpuassert.cuh $(INC)/gbd_pcg_compat/utils.cuh
SHIM_STEPS := $(INC)/mpcgpu_compat/linsys_steps.cuh
SHIM_SIM   := $(INC)/mpcsim.cuh $(INC)/pcg/sqp.cuh $(INC)/qdldl/sqp.cuh $(INC)/mpcgpu_compat/sqp_stages.cuh $(SHIM_STEPS) $(SHIM_PCG)

EXAMPLES := examples/sqp_pcg_callsite examples/sqp_pcg_callsite_f64 examples/sqp_pcg_callsite_f64_n128 examples/sqp_linsys_chain examples/sqp_linsys_chain_f64 \
            examples/mpcsim_shim_demo_pcg examples/mpcsim_shim_demo_qdldl examples/mpcsim_iiwa_demo_pcg examples/mpcsim_iiwa_demo_qdldl \
            examples/bd_utils_probe examples/multi_gpu_pcg

.PHONY: all lib examples oracle clean
all: lib examples
lib: $(LIB)
examples: $(EXAMPLES)
oracle:
	$(MAKE) -C oracle libmpcg_oracle.so

# ---- the library: one object per translation unit of csrc/, one link ----
$(CSRC)/%.o: $(CSRC)/%.hip $(LIB_DEPS)
	$(HIPCC) $(LIBFLAGS) -c $< -o $@
$(LIB): $(LIB_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(LIB_OBJS) -o $@

# ---- C++ call-site programs over the shim headers (the reference's include paths) ----
examples/sqp_pcg_callsite: examples/sqp_pcg_callsite.cpp $(LIB) $(SHIM_PCG)
	$(HIPCC) $(EXFLAGS) -I$(INC)/gbd_pcg_compat $< $(LINKLIB) -lpthread -o $@
examples/sqp_pcg_callsite_f64: examples/sqp_pcg_callsite.cpp $(LIB) $(SHIM_PCG)
	$(HIPCC) $(EXFLAGS) -DUSE_DOUBLES -I$(INC)/gbd_pcg_compat $< $(LINKLIB) -lpthread -o $@
examples/sqp_pcg_callsite_f64_n128: examples/sqp_pcg_callsite.cpp $(LIB) $(SHIM_PCG)
	$(HIPCC) $(EXFLAGS) -DUSE_DOUBLES -DKNOT_POINTS=128 -I$(INC)/gbd_pcg_compat $< $(LINKLIB) -lpthread -o $@
examples/sqp_linsys_chain: examples/sqp_linsys_chain.cpp $(LIB) $(SHIM_PCG) $(SHIM_STEPS)
	$(HIPCC) $(EXFLAGS) -I$(INC)/gbd_pcg_compat -I$(INC)/mpcgpu_compat $< $(LINKLIB) -o $@
examples/sqp_linsys_chain_f64: examples/sqp_linsys_chain.cpp $(LIB) $(SHIM_PCG) $(SHIM_STEPS)
	$(HIPCC) $(EXFLAGS) -DUSE_DOUBLES -I$(INC)/gbd_pcg_compat -I$(INC)/mpcgpu_compat $< $(LINKLIB) -o $@
examples/mpcsim_shim_demo_pcg: examples/mpcsim_shim_demo.cpp $(LIB) $(SHIM_SIM)
	$(HIPCC) $(EXFLAGS) -DLINSYS_SOLVE=1 -I$(INC) $< $(LINKLIB) -o $@
examples/mpcsim_shim_demo_qdldl: examples/mpcsim_shim_demo.cpp $(LIB) $(SHIM_SIM)
	$(HIPCC) $(EXFLAGS) -DLINSYS_SOLVE=0 -I$(INC) $< $(LINKLIB) -o $@
examples/mpcsim_iiwa_demo_pcg: examples/mpcsim_iiwa_demo.cpp $(LIB) $(SHIM_SIM)
	$(HIPCC) $(EXFLAGS) -DLINSYS_SOLVE=1 -I$(INC) $< $(LINKLIB) -o $@
examples/mpcsim_iiwa_demo_qdldl: examples/mpcsim_iiwa_demo.cpp $(LIB) $(SHIM_SIM)
	$(HIPCC) $(EXFLAGS) -DLINSYS_SOLVE=0 -I$(INC) $< $(LINKLIB) -o $@
examples/bd_utils_probe: examples/bd_utils_probe.cpp $(INC)/gbd_pcg_compat/utils.cuh
	$(HIPCC) $(EXFLAGS) -I$(INC)/gbd_pcg_compat $< -o $@
examples/multi_gpu_pcg: examples/multi_gpu_pcg.cpp $(LIB) $(INC)/mpcg.h
	$(HIPCC) $(EXFLAGS) $< $(LINKLIB) -lrccl -lpthread -o $@

clean:
	rm -f $(LIB) $(LIB_OBJS) $(EXAMPLES)
	$(MAKE) -C oracle clean
