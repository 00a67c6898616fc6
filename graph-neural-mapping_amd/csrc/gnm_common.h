// Shared device/host helpers for the gfx950 GIN hot-path kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GNM_OK 0
#define GNM_ERR_BAD_ARG (-1)
#define GNM_ERR_UNSUPPORTED (-2)

#define GNM_CHECK_LAUNCH()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

#define GNM_HIP(call)                               \
    do {                                            \
        hipError_t e__ = (call);                    \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static constexpr int kWave = 64;           // CDNA wavefront
static constexpr int kLdsBudget = 160 * 1024;  // bytes of LDS per CU (gfx950)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
