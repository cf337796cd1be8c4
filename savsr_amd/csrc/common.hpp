// Shared helpers for libsavsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/savsr_hip.h"

namespace savsr {

void set_error(const char* fmt, ...);

inline int fail_arg(const char* what) {
    set_error("invalid argument: %s", what);
    return SAVSR_E_ARG;
}

// hipFuncAttributeMaxDynamicSharedMemorySize for kernels that use more than 64 KiB of dynamic LDS.  The attribute is
// per (function, DEVICE): the guard records the size set per (function, current device), so a process driving several GPUs
// (one engine per device, or the reference's non-distributed DataParallel flow) sets it on each of them, and a later call
// asking for more LDS sets it again.  Thread-safe (mutex around the table).
int ensure_dynamic_lds(const void* fn, int bytes, const char* what);

int conv_prepare_device();     // conv_mfma.hip
int satu_prepare_device();     // satu.hip

inline int check_launch(const char* kernel) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", kernel, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// conv tile / packed weight-image geometry (shared by conv_mfma.hip and osconv.hip; mirrored in engine.py)
constexpr int CONV_TH = 8;     // pixel rows per block (one per wave)
constexpr int CONV_TW = 32;    // pixel columns per block (= MFMA N)
__host__ __device__ constexpr int conv_kc(int ksize) { return ksize == 3 ? 16 : 32; }   // input channels per K phase
__host__ __device__ inline int conv_cot(int cout) { return cout > 32 ? 64 : 32; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Row of a 32x32 MFMA accumulator register: C/D layout col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)   (cdna_hip_programming.md section 3).
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace savsr
