// gpuassert.cuh — HIP-native stand-in for the header of the same name in the reference's GBD-PCG
// submodule (included at include/pcg/sqp.cuh:18 via gpu_pcg.cuh, include/pcg/linsys_setup.cuh:3).
// Error convention of the reference: print and abort the process.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define gpuErrchk(ans) { gpuAssert((ans), __FILE__, __LINE__); }
inline void gpuAssert(hipError_t code, const char* file, int line, bool abort = true) {
    if (code != hipSuccess) {
        fprintf(stderr, "GPUassert: %s %s %d\n", hipGetErrorString(code), file, line);
        if (abort) exit(code);
    }
}
