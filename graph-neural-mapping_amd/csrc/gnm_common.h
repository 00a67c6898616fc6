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

// ---- host-side launch helpers ------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a process that
// drives several GPUs must set it on each of them.  One DeviceOnce per kernel instantiation keeps
// a flag per device ordinal (thread-safe; a racing second setter only repeats an idempotent call).
#include "gnm_once.h"
#define GNM_ALLOW_FULL_LDS(kernel_ptr)                                                               \
    do {                                                                                             \
        static GnmDeviceOnce once__;                                                                 \
        int dev__ = 0;                                                                               \
        GNM_HIP(hipGetDevice(&dev__));                                                               \
        if (once__.first_use(dev__)) {                                                               \
            GNM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr),                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget)); \
            once__.mark(dev__);                                                                      \
        }                                                                                            \
    } while (0)

// Tuning knobs are read from the environment ONCE per process (function-local statics at the call sites).
static inline int gnm_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static constexpr int kWave = 64;           // CDNA wavefront
static constexpr int kLdsBudget = 160 * 1024;  // bytes of LDS per CU (gfx950)

// ReLU the way torch computes it (graphcnn.py:166 / :190, mlp.py:48): a NaN stays a NaN, where fmaxf(x, 0) would
// return 0 (IEEE maxNum) and quietly repair a poisoned row -- e.g. the 0/0 row of an isolated node under neighbour
// "average" + learn_eps (graphcnn.py:157-158), which in the reference makes that graph's readout and logits NaN.
// One instruction on gfx950 (v_maximum3_f32, IEEE-754-2019 maximum).
__device__ __forceinline__ float gnm_relu(float x) { return __builtin_elementwise_maximum(x, 0.f); }

// An entry of the inverse Infomax permutation (graphcnn.py:198-201; scattered by the discriminator kernels from the
// caller's permutation).  The HOST validates every permutation it uploads (gnm/core.py perm_to_device: a permutation
// of 0 .. B-1 or GnmError), so the entry is used as it is; round 3's clamp into [0, B) -- which turned a bad
// permutation into silently wrong gradients instead of a fault -- survives only for debugging builds.
__device__ __forceinline__ int gnm_perm_entry(int v, int B) {
#ifdef GNM_DEBUG_CLAMP_PERM
    return min(max(v, 0), B - 1);
#else
    (void)B;
    return v;
#endif
}

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
