// GIN neighbour aggregation on the matrix cores, TRANSPOSED roles (round 3 experiment -> see DESIGN.md): the same product
// as aggm.hip (torch.spmm(Adj_block, h) + the (1 + eps) self term, /root/reference models/graphcnn.py:154-161, :178-182)
// computed as  Y^T = H^T x Adj^T  instead of  Y = Adj x H.
//
// Why: aggm.hip stages the feature tile in LDS as three bf16 planes (77 KB) so that all row blocks can read it as the
// MFMA B operand.  Two workgroups fit a CU and each runs load -> split -> barrier -> product -> epilogue in sequence; the
// in-kernel timelines put matrix pipe, LDS and HBM at about a third busy each -- the kernel is bound by that phase
// structure ("what was tried").  With H as the A operand the fragment a lane needs is "eight consecutive rows of
// one column": exactly what a 4-byte load with lane = column delivers (128 contiguous bytes per half-wave), so the tile
// never goes through LDS, there is no barrier, and a wave is an independent stream: request rows k+2, split rows k in
// registers, multiply.  The price is that every wave of a unit loads and splits the whole [n, 32] tile itself, so a wave
// takes FOUR output row blocks (accumulators 4 x 16 registers) to spread that over 12 MFMAs per step; the waves of a
// workgroup (4: row blocks w, w + 4, w + 8, w + 12) read the same rows at about the same time (L1 / L2 hits).
// The adjacency bits are the B operand (lane = output row, its 8 bits of the step expanded through the 16-entry table, as
// in aggm.hip).  Accumulators come out as lane = output row, 16 columns in registers: rows are stored as 16-byte pieces.
//
// Forms: "sum" neighbour pooling, F a multiple of 32, plain or with the forward prologue (BatchNorm + ReLU of the
// previous layer on the way in, the activation and its graph readout written by one wave of the unit).  Everything else
// stays on aggm.hip / agg.hip.
#include "gnm_agg_args.h"
#include <string.h>
#include <type_traits>

typedef __bf16 t_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int t_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int t_u32x2 __attribute__((ext_vector_type(2)));

static constexpr int kAggtWaves = 4;
static constexpr int kAggtMaxN = 416;            // 13 row blocks, 26 steps

__device__ __forceinline__ unsigned aggt_pair_hi(unsigned lo_word, unsigned hi_word) {
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);
}

// eight floats (consecutive rows of one column) -> the three exact bf16x8 planes (see gnm_lin_split_kernel)
__device__ __forceinline__ void aggt_split8(const float* f, t_bf16x8& p1, t_bf16x8& p2, t_bf16x8& p3) {
    unsigned a1[8], a2[8], a3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a1[j] = __float_as_uint(f[j]) & 0xFFFF0000u;
        const float r1 = f[j] - __uint_as_float(a1[j]);
        a2[j] = __float_as_uint(r1) & 0xFFFF0000u;
        a3[j] = __float_as_uint(r1 - __uint_as_float(a2[j]));
    }
    t_u32x4 q1, q2, q3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q1[j] = aggt_pair_hi(a1[2 * j], a1[2 * j + 1]);
        q2[j] = aggt_pair_hi(a2[2 * j], a2[2 * j + 1]);
        q3[j] = aggt_pair_hi(a3[2 * j], a3[2 * j + 1]);
    }
    p1 = __builtin_bit_cast(t_bf16x8, q1); p2 = __builtin_bit_cast(t_bf16x8, q2); p3 = __builtin_bit_cast(t_bf16x8, q3);
}

__global__ void __launch_bounds__(kAggtWaves * 64, 3) gnm_aggt_kernel(const AggArgs p) {
    __shared__ __attribute__((aligned(16))) char ring[8 * 3072];     // eight steps of operand planes (two groups of four)
    __shared__ __attribute__((aligned(16))) char lut[128];
    __shared__ float rsum[kAggtWaves][64];
    const int nc = p.F >> 5;
    const int grp = blockIdx.x / (8 * nc), within = blockIdx.x - grp * (8 * nc);
    const int b = grp * 8 + (within & 7);
    const int cb = within >> 3;
    if (b >= p.n_graphs) return;
    const int row0 = p.node_off[b];
    const int n = p.node_off[b + 1] - row0;
    const int col0 = cb * 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const bool pro = p.p_scale != nullptr;
    if (n <= 0) {
        if (pro && p.p_gf && tid < 32) p.p_gf[(size_t)b * p.p_ldgf + col0 + tid] = p.p_gf_avg ? 0.f / 0.f : 0.f;
        return;
    }
    const int W = (n + 31) >> 5;
    const int ksteps = (n + 15) >> 4;
    if (tid < 16) {          // nibble e -> bf16 (bit 0, bit 1, bit 2, bit 3) as two words
        const unsigned one = 0x3F80u;
        t_u32x2 v;
        v.x = ((tid & 1) ? one : 0u) | ((tid & 2) ? one << 16 : 0u);
        v.y = ((tid & 4) ? one : 0u) | ((tid & 8) ? one << 16 : 0u);
        *reinterpret_cast<t_u32x2*>(lut + 8 * tid) = v;
    }
    // this wave's output row blocks: role, role + 4, role + 8, role + 12 (rotated with the column block)
    const int role = (wave + cb) & 3;
    const int HPW = (((W + 1) >> 1) + 3) & ~3;                   // words per half row of the bit adjacency (4 or 8)
    unsigned pk[4][8];
    {
        const uint32_t* gbits = p.adj_bits + p.b_bits_off[b];
        const t_u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rb = min(role + 4 * q, W - 1);
            const t_u32x4* rp = reinterpret_cast<const t_u32x4*>(gbits + (size_t)(rb * 32 + i) * (2 * HPW) + h * HPW);
            const t_u32x4 a0 = rp[0];
            const t_u32x4 a1 = HPW > 4 ? rp[1] : z4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { pk[q][j] = a0[j]; pk[q][4 + j] = a1[j]; }
        }
    }
    const int nblk = role + 12 < W ? 4 : (role + 8 < W ? 3 : (role + 4 < W ? 2 : (role < W ? 1 : 0)));   // wave-uniform
    // the tile, lane = column: H[16 s + 8 h + j][col0 + i]; rows past n read zero (descriptor)
    const unsigned xbytes = (unsigned)(((size_t)(n - 1) * p.ldx + 32) * 4);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (size_t)row0 * p.ldx + col0, 0, (int)xbytes, 0x00020000);
    const int xvo = (8 * h * p.ldx + i) * 4;                      // + (16 s + j) rows in the scalar offset
    const int xrow = p.ldx * 4;
    float psc = 1.f, psh = 0.f;
    if (pro) { psc = p.p_scale[col0 + i]; psh = p.p_shift[col0 + i]; }
    // every wave produces the steps s = 4 g + wave of the ring (below) and writes the activation rows of those steps
    const unsigned hbytes = (pro && p.p_hout) ? (unsigned)(((size_t)(n - 1) * p.p_ldh + 32) * 4) : 0u;
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
        (p.p_hout ? p.p_hout : p.y) + (size_t)row0 * (p.p_hout ? p.p_ldh : p.ldy) + col0, 0, (int)hbytes, 0x00020000);
    const int hvo = (8 * h * p.p_ldh + i) * 4;
    const int hrow = p.p_ldh * 4;
    float csum = 0.f;

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    auto bfrag = [&](unsigned pkw, int m) -> t_bf16x8 {
        const unsigned byte3 = m == 0 ? (pkw << 3) : (pkw >> (8 * m - 3));
        const unsigned lo = byte3 & 0x78u, hi = (byte3 >> 4) & 0x78u;
        const t_u32x2 l2 = *reinterpret_cast<const t_u32x2*>(lut + lo);
        const t_u32x2 h2 = *reinterpret_cast<const t_u32x2*>(lut + hi);
        const t_u32x4 q = {l2.x, l2.y, h2.x, h2.y};
        return __builtin_bit_cast(t_bf16x8, q);
    };
    auto request = [&](float (&d)[8], int s) {                    // s past the graph: offsets past the descriptor, zeros
#ifdef AGGT_EXP_NO_LOAD       // timing experiments only (wrong results)
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = __int_as_float(xvo + s + j);
#else
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xvo, (16 * s + j) * xrow, 0));
#endif
    };
    // wave w turns step 4 g + w of group g into the three operand planes and leaves them in the ring (slot = step & 7,
    // plane-major inside a slot, 16 bytes per lane: lane-linear writes and reads)
    auto produce = [&](const float (&d)[8], int s) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = d[j];
        if (pro) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f[j] = gnm_relu(f[j] * psc + psh);
                if (16 * s + 8 * h + j >= n) f[j] = 0.f;          // (a clipped row read zero and the affine map moved it)
                csum += f[j];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(f[j]), rh, hvo, (16 * s + j) * hrow, 0);
            }
        }
        t_bf16x8 a1, a2, a3;
        aggt_split8(f, a1, a2, a3);
        char* slot = ring + (s & 7) * 3072 + lane * 16;
        *reinterpret_cast<t_bf16x8*>(slot) = a1;
        *reinterpret_cast<t_bf16x8*>(slot + 1024) = a2;
        *reinterpret_cast<t_bf16x8*>(slot + 2048) = a3;
    };
    float hb[2][8];                                               // this wave's steps of the next two groups
    request(hb[0], wave);
    request(hb[1], 4 + wave);
    if (wave < ksteps) produce(hb[0], wave);                      // group 0
    __syncthreads();                                              // (also the table)
    // the group loop, instantiated per number of row blocks of the wave (a run-time guard per MFMA left ~670 branches
    // in the unrolled body and nothing for the scheduler to move)
    auto groups = [&](auto nb_tag) {
        constexpr int NB = decltype(nb_tag)::value;
    #pragma unroll
        for (int g = 0; g < 7; ++g) {
            if (4 * g < ksteps) {                                     // workgroup-uniform
                if (4 * (g + 2) < ksteps) request(hb[g & 1], 4 * (g + 2) + wave);
                // group g + 1 into the other half of the ring (its loads were requested one iteration ago)
                if (4 * (g + 1) + wave < ksteps) produce(hb[(g + 1) & 1], 4 * (g + 1) + wave);
                auto step = [&](int s) {
                    const char* slot = ring + (s & 7) * 3072 + lane * 16;
                    const t_bf16x8 a1 = *reinterpret_cast<const t_bf16x8*>(slot);
                    const t_bf16x8 a2 = *reinterpret_cast<const t_bf16x8*>(slot + 1024);
                    const t_bf16x8 a3 = *reinterpret_cast<const t_bf16x8*>(slot + 2048);
                    // plane-major issue order: consecutive MFMAs go to different accumulators (a dependent one issued
                    // right behind its predecessor waits out the pipeline)
                    t_bf16x8 bq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = bfrag(pk[q < NB ? q : 0][s >> 2], s & 3);
#ifdef AGGT_EXP_NO_MFMA
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q][s & 15] += (float)a3[0] + (float)a2[1] + (float)a1[2] + (float)bq[q][q];
#else
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q < NB) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bq[q], acc[q], 0, 0, 0);   // small planes first
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q < NB) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bq[q], acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q < NB) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[q], acc[q], 0, 0, 0);
#endif
                };
                // (the four steps as ONE basic block -- no per-step guard -- let the scheduler hoist every LDS read of the
                //  group: 381 spilled registers at the 168 this kernel may use)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (4 * g + u < ksteps) step(4 * g + u);      // workgroup-uniform
                __syncthreads();        // group g + 1 is complete; group g has been read (its half is written two groups on)
            }
        }
    };
    switch (nblk) {                                               // wave-uniform
        case 4: groups(std::integral_constant<int, 4>{}); break;
        case 3: groups(std::integral_constant<int, 3>{}); break;
        case 2: groups(std::integral_constant<int, 2>{}); break;
        case 1: groups(std::integral_constant<int, 1>{}); break;
        default: groups(std::integral_constant<int, 0>{}); break;
    }
    // ---- epilogue: lane = output row 32 rb + i, columns 8 g + 4 h + 0..3 of the block in acc[q][4 g ..] ----------------
    const float selfw = p.self_loop ? 1.f : (p.eps ? 1.f + *p.eps : 1.f);
    const unsigned ybytes = (unsigned)(((size_t)(n - 1) * p.ldy + p.F) * 4);
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)row0 * p.ldy, 0, (int)ybytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < nblk) {
            const int vrow = (role + 4 * q) * 32 + i;
            const float* xr = p.x + (size_t)(row0 + min(vrow, n - 1)) * p.ldx + col0 + 4 * h;
            float4 hs[4];
#ifdef AGGT_EXP_NO_SELF
#pragma unroll
            for (int g = 0; g < 4; ++g) hs[g] = make_float4(1.f, 2.f, 3.f, (float)vrow);
#else
#pragma unroll
            for (int g = 0; g < 4; ++g) hs[g] = *reinterpret_cast<const float4*>(xr + 8 * g);
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = hs[g];
                if (pro) {
                    const float4 sc = *reinterpret_cast<const float4*>(p.p_scale + col0 + 8 * g + 4 * h);
                    const float4 sh = *reinterpret_cast<const float4*>(p.p_shift + col0 + 8 * g + 4 * h);
                    v.x = gnm_relu(v.x * sc.x + sh.x); v.y = gnm_relu(v.y * sc.y + sh.y);
                    v.z = gnm_relu(v.z * sc.z + sh.z); v.w = gnm_relu(v.w * sc.w + sh.w);
                }
                t_u32x4 o;
                o[0] = __float_as_uint(acc[q][4 * g + 0] + selfw * v.x);
                o[1] = __float_as_uint(acc[q][4 * g + 1] + selfw * v.y);
                o[2] = __float_as_uint(acc[q][4 * g + 2] + selfw * v.z);
                o[3] = __float_as_uint(acc[q][4 * g + 3] + selfw * v.w);
#ifdef AGGT_EXP_NO_STORE
                if (o[0] == 0x12345678u && o[3] == 0x9abcdef0u)
#endif
                __builtin_amdgcn_raw_buffer_store_b128(o, ry, (unsigned)((vrow * p.ldy + col0 + 8 * g + 4 * h) * 4), 0, 0);
            }
        }
    }
    if (pro && p.p_gf) {                                          // the four waves' shares of the rows, fixed order
        rsum[wave][lane] = csum;
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < kAggtWaves; ++w) t += rsum[w][tid] + rsum[w][32 + tid];
            p.p_gf[(size_t)b * p.p_ldgf + col0 + tid] = p.p_gf_avg ? t * (1.f / (float)n) : t;
        }
    }
}

// The forms this kernel takes (the caller falls back to aggm.hip otherwise).
static bool aggt_shape_ok(const AggArgs& a, int n_max) {
    if (!a.adj_bits || !a.b_bits_off || (reinterpret_cast<uintptr_t>(a.adj_bits) & 15)) return false;
    if (n_max < 1 || n_max > kAggtMaxN) return false;
    if (a.F < 32 || (a.F & 31) || a.F > 256) return false;
    if (a.average || a.sZ || a.hfwd || a.deps_partial || !a.y) return false;
    if ((a.ldx & 3) || (a.ldy & 3) || ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y)) & 15)) return false;
    if (a.p_scale && ((reinterpret_cast<uintptr_t>(a.p_scale) | reinterpret_cast<uintptr_t>(a.p_shift)) & 15)) return false;
    if ((long long)(n_max + 48) * (a.ldx > a.ldy ? a.ldx : a.ldy) * 4 >= (1LL << 31)) return false;
    return true;
}

extern "C" int gnm_aggt_launch(const AggArgs* args, int B, int n_max, void* stream) {
    AggArgs a = *args;
    if (!aggt_shape_ok(a, n_max)) return GNM_ERR_UNSUPPORTED;
    a.n_graphs = B;
    a.n16_max = ((n_max + 15) / 16) * 16;
    a.stamps = nullptr;
    const int nc = a.F / 32;
    const int grid = ((B + 7) / 8) * 8 * nc;
    hipLaunchKernelGGL(gnm_aggt_kernel, dim3(grid), dim3(kAggtWaves * 64), 0, reinterpret_cast<hipStream_t>(stream), a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// gnm_aggm / gnm_aggm_fwd_bnrelu with this kernel (same arguments); GNM_ERR_UNSUPPORTED for forms it does not take.
extern "C" int gnm_aggt(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off, const int64_t* b_col_off,
                        const uint32_t* adj_bits, const int64_t* b_bits_off, const int32_t* deg_rowptr,
                        const int64_t* b_deg_off, const int32_t* node_off, int B, int n_max, const float* x, int ldx,
                        float* y, int ldy, int F, const float* eps, int average, int self_loop, int backward,
                        const float* hfwd, int ldh, double* deps_partial, void* stream) {
    if (B <= 0) return GNM_OK;
    if (F <= 0 || n_max < 0) return GNM_ERR_BAD_ARG;
    AggArgs a;
    memset(&a, 0, sizeof(a));
    a.rowptr = rowptr; a.col = col; a.b_rp_off = b_rp_off; a.b_col_off = b_col_off;
    a.adj_bits = adj_bits; a.b_bits_off = b_bits_off;
    a.deg_rowptr = deg_rowptr ? deg_rowptr : rowptr;
    a.b_deg_off = b_deg_off ? b_deg_off : b_rp_off;
    a.node_off = node_off; a.x = x; a.y = y; a.eps = eps; a.hfwd = hfwd; a.deps_partial = deps_partial;
    a.ldx = ldx; a.ldy = ldy; a.ldh = ldh; a.F = F; a.nslices = 1;
    a.average = average; a.self_loop = self_loop; a.backward = backward;
    return gnm_aggt_launch(&a, B, n_max, stream);
}

extern "C" int gnm_aggt_fwd_bnrelu(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off,
                                   const int64_t* b_col_off, const uint32_t* adj_bits, const int64_t* b_bits_off,
                                   const int32_t* node_off, int B, int n_max, const float* z, int ldz,
                                   const float* scale, const float* shift, float* hout, int ldh, float* gf, int ldgf,
                                   int graph_avg, float* y, int ldy, int F, const float* eps, int average,
                                   int self_loop, void* stream) {
    if (B <= 0) return GNM_OK;
    if (!y || !z || !scale || !shift) return GNM_ERR_UNSUPPORTED;
    AggArgs a;
    memset(&a, 0, sizeof(a));
    a.rowptr = rowptr; a.col = col; a.b_rp_off = b_rp_off; a.b_col_off = b_col_off;
    a.adj_bits = adj_bits; a.b_bits_off = b_bits_off;
    a.deg_rowptr = rowptr; a.b_deg_off = b_rp_off;
    a.node_off = node_off; a.x = z; a.y = y; a.eps = eps;
    a.ldx = ldz; a.ldy = ldy; a.F = F; a.nslices = 1;
    a.average = average; a.self_loop = self_loop; a.backward = 0;
    a.p_scale = scale; a.p_shift = shift; a.p_hout = hout; a.p_gf = gf; a.p_ldh = ldh; a.p_ldgf = ldgf;
    a.p_gf_avg = graph_avg;
    return gnm_aggt_launch(&a, B, n_max, stream);
}
