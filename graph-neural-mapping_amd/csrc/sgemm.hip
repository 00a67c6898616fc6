// The three [B, L*H]-sized matrix products of the Infomax tail (/root/reference models/discriminator.py:30-31 through
// nn.Bilinear, restructured in gnm/core.py as  U = sigmoid(g_f) Wd^T  and its backward  dWd = dU^T sigmoid(g_f),
// T = dU Wd) as ONE hand-written kernel instead of three hipBLASLt calls.
//
// These products are small (2 x 1024 x 320 x 320 flop at the headline batch): what they cost is latency, not throughput.
// A workgroup of eight (K > 512: sixteen) waves owns one 32 x 32 output tile and splits the contraction index among its waves (steps of 16,
// interleaved); both operand fragments come straight from global memory in MFMA operand order -- a row-major operand as
// two 16-byte loads of eight consecutive k per lane, a column-major one as eight 4-byte loads with lane = column (see
// gnm_wgrad_split128_kernel) -- are split in registers into three exact bf16 planes and multiplied as six bf16 terms
// (see gnm_lin_split_kernel: fp32-accurate, 16x the rate of the fp32 instruction).  The eight partial tiles are added in a
// fixed order through LDS: one launch, bitwise reproducible.  320 or 100 workgroups -- about one per CU.
#include "gnm_common.h"

typedef __bf16 sg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int sg_u32x4 __attribute__((ext_vector_type(4)));

struct SgArgs {
    const float* A; const float* B; float* C;
    int lda, ldb, ldc;
    int M, N, K;
    int a_cols;      // 0: A[m][k] row-major (lda >= K); 1: A given as [k][m] (lda >= M), lane = column loads
    int b_cols;      // 0: B given as [n][k] (ldb >= K: C = A Bn^T); 1: B given as [k][n] (ldb >= N)
    int a_vec, b_vec;   // row-major operand 16-byte addressable (else 4-byte loads)
};

__device__ __forceinline__ void sg_split8(const float* f, sg_bf16x8& p1, sg_bf16x8& p2, sg_bf16x8& p3) {
    unsigned a1[8], a2[8], a3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a1[j] = __float_as_uint(f[j]) & 0xFFFF0000u;
        const float r1 = f[j] - __uint_as_float(a1[j]);
        a2[j] = __float_as_uint(r1) & 0xFFFF0000u;
        a3[j] = __float_as_uint(r1 - __uint_as_float(a2[j]));
    }
    sg_u32x4 q1, q2, q3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q1[j] = __builtin_amdgcn_perm(a1[2 * j + 1], a1[2 * j], 0x07060302u);
        q2[j] = __builtin_amdgcn_perm(a2[2 * j + 1], a2[2 * j], 0x07060302u);
        q3[j] = __builtin_amdgcn_perm(a3[2 * j + 1], a3[2 * j], 0x07060302u);
    }
    p1 = __builtin_bit_cast(sg_bf16x8, q1); p2 = __builtin_bit_cast(sg_bf16x8, q2); p3 = __builtin_bit_cast(sg_bf16x8, q3);
}

// One operand fragment of step s for this lane: eight values along the contraction index (k = 16 s + 8 h + 0..7) of
// row / column `line` (= 32 tile + i).  Rows, columns and steps past the operand read zero (offsets the descriptor
// clips).  MODE 0: source [line][k], 16-byte addressable and K a multiple of 8 (two 16-byte loads); 1: source [line][k],
// 4-byte loads; 2: source [k][line], lane = column.  Branch-free: with a run-time choice between the load forms inside the
// unrolled request loop the K = 320 products took 15 us.
template <int MODE>
__device__ __forceinline__ void sg_fetch(const __amdgpu_buffer_rsrc_t rs, int ld, int line, int nlines, int K, int s, int h,
                                         float (&f)[8]) {
    const int k0 = 16 * s + 8 * h;
    if constexpr (MODE == 2) {
        // rows past K lie past the end of the operand: the descriptor's range check (which covers the scalar offset on
        // gfx950, see linear.hip gnm_tile_rsrc) returns zero for them -- one vector offset per fragment, no per-element selects
        const unsigned voff = line < nlines ? (unsigned)((8 * h * ld + line) * 4) : 0x80000000u;   // (past any operand; no wrap with the row offset)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            f[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (16 * s + j) * ld * 4, 0));
    } else if constexpr (MODE == 0) {
        const unsigned off = (line < nlines && k0 < K) ? (unsigned)((line * ld + k0) * 4) : 0xFFFFFFE0u;
        const sg_u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        const sg_u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 16, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = __uint_as_float(v0[j]); f[4 + j] = __uint_as_float(v1[j]); }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = (line < nlines && k0 + j < K) ? (unsigned)((line * ld + k0 + j) * 4) : 0xFFFFFFF0u;
            f[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
        }
    }
}

// kSgWaves waves of a workgroup split the contraction index (8; 16 when K > 512: a wave then still has at most four steps)
template <int AMODE, int BMODE, int kSgWaves>
__global__ void __launch_bounds__(kSgWaves * 64) gnm_small_gemm_kernel(const SgArgs p) {
    __shared__ float part[kSgWaves][16][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int ntn = (p.N + 31) >> 5;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int row = 32 * tm + i;              // A line of this lane (output row m)
    const int col = 32 * tn + i;              // B line of this lane (output column n)
    // descriptors over the whole operands (byte sizes fit 32 bits: checked by the launcher)
    constexpr bool ACOLS = AMODE == 2, BCOLS = BMODE == 2;
    const unsigned abytes = (unsigned)(((size_t)((ACOLS ? p.K : p.M) - 1) * p.lda + (ACOLS ? p.M : p.K)) * 4);
    const unsigned bbytes = (unsigned)(((size_t)((BCOLS ? p.K : p.N) - 1) * p.ldb + (BCOLS ? p.N : p.K)) * 4);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, (int)bbytes, 0x00020000);
    const int nsteps = (p.K + 15) >> 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // this wave's steps: wave, wave + 8, ...  These products are latency-bound (a wave has two to eight steps): four
    // steps' operands are requested at once, then split and multiplied (one step ahead only, the K = 320 products took
    // 15 us and the K = 1024 one 54 us against hipBLASLt's 9.5-12.4 us)
    constexpr int D = 4;
    float fa[D][8], fb[D][8];
    for (int base = wave; base < nsteps; base += kSgWaves * D) {  // (wave-uniform trip count)
#pragma unroll
        for (int u = 0; u < D; ++u) {                             // (steps past K read zeros)
            sg_fetch<AMODE>(ra, p.lda, row, p.M, p.K, base + kSgWaves * u, h, fa[u]);
            sg_fetch<BMODE>(rb, p.ldb, col, p.N, p.K, base + kSgWaves * u, h, fb[u]);
        }
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (base + kSgWaves * u < nsteps) {                   // wave-uniform
                sg_bf16x8 a1, a2, a3, b1, b2, b3;
                sg_split8(fa[u], a1, a2, a3);
                sg_split8(fb[u], b1, b2, b3);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);      // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
    __syncthreads();
    // accumulator element (r, lane): row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
    for (int e = tid; e < 16 * 64; e += kSgWaves * 64) {
        const int r = e >> 6, ln = e & 63;
        float v = part[0][r][ln];
#pragma unroll
        for (int w = 1; w < kSgWaves; ++w) v += part[w][r][ln];
        const int m = 32 * tm + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int n = 32 * tn + (ln & 31);
        if (m < p.M && n < p.N) p.C[(size_t)m * p.ldc + n] = v;
    }
}

// C[M,N] = A' B'  with A' = A (a_cols = 0, A is [M][K]) or A^T (a_cols = 1, A is [K][M]) and B' = B^T (b_cols = 0, B is
// [N][K]) or B (b_cols = 1, B is [K][N]); fp32 in and out, fp32-accurate (split-precision bf16 products), fixed summation
// order.  GNM_ERR_UNSUPPORTED (nothing launched) for operands larger than a 32-bit byte offset.
extern "C" int gnm_small_gemm(const float* A, int lda, int a_cols, const float* B, int ldb, int b_cols, float* C, int ldc,
                              int M, int N, int K, void* stream) {
    if (M <= 0 || N <= 0) return GNM_OK;
    if (K <= 0 || !A || !B || !C) return GNM_ERR_BAD_ARG;
    const long long arows = a_cols ? K : M, brows = b_cols ? K : N;
    if (arows * (long long)lda * 4 >= (1LL << 31) || brows * (long long)ldb * 4 >= (1LL << 31)) return GNM_ERR_UNSUPPORTED;
    SgArgs a;
    a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.a_cols = a_cols; a.b_cols = b_cols;
    const int amode = a_cols ? 2 : (((lda & 3) == 0 && (K & 7) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0) ? 0 : 1);
    const int bmode = b_cols ? 2 : (((ldb & 3) == 0 && (K & 7) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0) ? 0 : 1);
    a.a_vec = amode == 0; a.b_vec = bmode == 0;
    const int grid = ((M + 31) / 32) * ((N + 31) / 32);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define GNM_SG_CASE(AM_, BM_)                                                                                   \
    if (amode == AM_ && bmode == BM_) {                                                                         \
        if (K > 512) hipLaunchKernelGGL((gnm_small_gemm_kernel<AM_, BM_, 16>), dim3(grid), dim3(1024), 0, s, a); \
        else hipLaunchKernelGGL((gnm_small_gemm_kernel<AM_, BM_, 8>), dim3(grid), dim3(512), 0, s, a);          \
    }
    GNM_SG_CASE(0, 0) GNM_SG_CASE(0, 1) GNM_SG_CASE(0, 2)
    GNM_SG_CASE(1, 0) GNM_SG_CASE(1, 1) GNM_SG_CASE(1, 2)
    GNM_SG_CASE(2, 0) GNM_SG_CASE(2, 1) GNM_SG_CASE(2, 2)
#undef GNM_SG_CASE
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}
