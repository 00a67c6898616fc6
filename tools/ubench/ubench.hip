// Instruction-cost micro-benchmarks for the aggregation inner loop (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench ; run: ./ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) const f32x4* lds_cf4p;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d line %d\n", (int)e, __LINE__); exit(1);} } while (0)

template <int S> __device__ __forceinline__ unsigned bc(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + S, 0xf, 0xf, false);
}

// MODE 0: v_pk_add_f32 x16/iter   1: v_add_f32 x32/iter   2: dpp add x16/iter
// MODE 3: ds_read_b128 x16/iter (dpp addr) + sum at the end of the 16
// MODE 4: full gather step mix: 16 x (dpp + ds_read_b128) + 32 pk_add
// MODE 5: as 4 but plain v_add_f32 (asm) instead of pk
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, const unsigned* idx, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* tile = (float4*)smem;
    for (int i = threadIdx.x; i < 401 * 16; i += blockDim.x) tile[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned subb = base + (threadIdx.x & 15) * 16;
    unsigned valb = idx[threadIdx.x] * 256;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float s0 = 0, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    unsigned u0 = valb, u1 = valb + 1, u2 = valb + 2, u3 = valb + 3;
    double d0 = 1.0, d1 = 2.0, d2 = 3.0, d3 = 4.0, dk = 1e-300;
    for (int it = 0; it < ((MODE == 6 || MODE == 7) ? 0 : iters); ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dk));
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(a0[0]));
            }
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u0 += bc<1>(valb); u1 += bc<2>(valb); u2 += bc<3>(valb); u3 += bc<4>(valb);
                asm volatile("" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            }
        } else {
#define RD(S) (*(lds_cf4p)(uintptr_t)(bc<S>(valb) + subb))
            const f32x4 r0 = RD(0), r1 = RD(1), r2 = RD(2), r3 = RD(3), r4 = RD(4), r5 = RD(5), r6 = RD(6), r7 = RD(7),
                        r8 = RD(8), r9 = RD(9), r10 = RD(10), r11 = RD(11), r12 = RD(12), r13 = RD(13), r14 = RD(14), r15 = RD(15);
            if constexpr (MODE == 3) {
                // keep all reads live, minimal VALU: xor-fold as ints is still VALU; just sum two of them
                a0 += r0; a1 += r15;
                asm volatile("" :: "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(r8), "v"(r9), "v"(r10),
                             "v"(r11), "v"(r12), "v"(r13), "v"(r14));
            } else {
                a0 += (r0 + r1) + (r2 + r3);
                a1 += (r4 + r5) + (r6 + r7);
                a2 += (r8 + r9) + (r10 + r11);
                a3 += (r12 + r13) + (r14 + r15);
            }
            valb ^= 256;   // change rows a little so nothing is hoisted
        }
    }
    if constexpr (MODE == 6 || MODE == 7) {
        constexpr int NB = MODE == 6 ? 8 : 16;
        f32x4 A[NB], Bf[NB];
#define RDX(S) (*(lds_cf4p)(uintptr_t)(bc<S>(valb) + subb))
#define FILL8(X, O) X[O+0] = RDX(0); X[O+1] = RDX(1); X[O+2] = RDX(2); X[O+3] = RDX(3); X[O+4] = RDX(4); X[O+5] = RDX(5); X[O+6] = RDX(6); X[O+7] = RDX(7);
#define FILL8B(X, O) X[O+0] = RDX(8); X[O+1] = RDX(9); X[O+2] = RDX(10); X[O+3] = RDX(11); X[O+4] = RDX(12); X[O+5] = RDX(13); X[O+6] = RDX(14); X[O+7] = RDX(15);
        if constexpr (NB == 8) { FILL8(A, 0) } else { FILL8(A, 0) FILL8B(A, 8) }
        const int n2 = (MODE == 6) ? iters : iters / 2;   // same number of steps as the 16/iter modes
        for (int it = 0; it < n2; ++it) {
            valb ^= 256;
            if constexpr (NB == 8) { FILL8B(Bf, 0) } else { FILL8(Bf, 0) FILL8B(Bf, 8) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NB; j += 4) { a0 += (A[j] + A[j+1]) + (A[j+2] + A[j+3]); }
            __builtin_amdgcn_sched_barrier(0);
            valb ^= 256;
            if constexpr (NB == 8) { FILL8(A, 0) } else { FILL8(A, 0) FILL8B(A, 8) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NB; j += 4) { a1 += (Bf[j] + Bf[j+1]) + (Bf[j+2] + Bf[j+3]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        a2 += A[0] + A[NB-1];
    }
    f32x4 t = a0 + a1 + a2 + a3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = t[0] + t[1] + t[2] + t[3] + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + (float)(u0 + u1 + u2 + u3) + (float)(d0 + d1 + d2 + d3);
}

template <int MODE>
void run(const char* name, int threads, int iters, double inst_per_iter, float* out, unsigned* idx) {
    CHECK(hipFuncSetAttribute((const void*)&k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const size_t lds = 401 * 256;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), lds, 0, out, idx, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), lds, 0, out, idx, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double waves_per_simd = threads / 64.0 / 4.0;
    // cycles (at 2.4 GHz nominal) per instruction per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * inst_per_iter * waves_per_simd);
    printf("%-34s threads %4d: %8.3f ms  -> %.2f cyc/inst/SIMD @2.4GHz (per-wave %.1f)\n", name, threads, ms, cyc, cyc * waves_per_simd);
}

int main() {
    float* out; unsigned* idx;
    CHECK(hipMalloc(&out, 256 * 1024 * 4)); CHECK(hipMalloc(&idx, 1024 * 4));
    unsigned h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (i * 37 + 11) % 400;
    CHECK(hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice));
    const int it = 20000;
    for (int th : {256, 512, 1024}) {
        run<0>("v_pk_add_f32 (16/iter)", th, it, 16, out, idx);
        run<1>("v_add_f32 (32/iter)", th, it, 32, out, idx);
        run<2>("v_add_u32_dpp newbcast (16/iter)", th, it, 16, out, idx);
        run<3>("ds_read_b128 (16/iter)", th, it, 16, out, idx);
        run<4>("gather mix per step (16/iter)", th, it, 16, out, idx);
        run<6>("gather mix, 8+8 double-buffered", th, it, 16, out, idx);
        run<7>("gather mix, 16+16 double-buffered", th, it, 16, out, idx);
    }
    return 0;
}
