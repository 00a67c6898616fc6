// GIN neighbour aggregation on gfx950 (MI355X): the K1+K2+K3 fusion of SURVEY.md
// section 2.2, replacing torch.spmm(Adj_block, h) (+ degree spmm, divide, (1+eps)h)
// at /root/reference models/graphcnn.py:154-161 and :178-182, and its autograd
// backward.
//
// Design (one workgroup = one graph x one feature slice):
//   * the block-diagonal matrix is never materialised: each graph keeps a private
//     CSR (int32 rowptr, uint16 graph-local column ids) in a device-resident arena;
//   * phase A streams the graph's [n, FS] feature tile from HBM into LDS once,
//     coalesced (16 B per lane);
//   * phase B gathers neighbour rows FROM LDS with ds_read_b128: LPR = FS/4 lanes
//     cover one row, so one wave-instruction reads 64/LPR neighbour rows (1 KiB),
//     the LDS peak of 256 B/clk/CU.  For LPR == 16 (FS = 64 floats = 256-B rows)
//     the four 16-lane quarters of a wave each take every 4th neighbour of the SAME
//     destination row; neighbour ids are fetched 64 at a time with one coalesced
//     load and handed to the quarters with a DPP row_newbcast folded into the
//     address VALU (v_or_b32_dpp), so a gather step is 1 VALU + 1 ds_read_b128 +
//     4 adds.  256-B rows make every 16-lane ds_read_b128 group hit 64 distinct
//     banks whatever the row ids are (bank = (addr/4) % 64), so there are no
//     bank conflicts by construction;
//   * the quarters' partial sums are combined with two wave shuffles, the
//     mean/(1+eps) epilogue is applied in registers and `pooled` is written once.
//
// One kernel serves forward and backward through the identity
//   y[v] = post[v] * ( sum_{u in N(v)} pre[u] x[u] + selfA * pre[v] x[v] ) + selfB * x[v]
//   forward : pre = 1,           post = 1/deg' (average) or 1
//   backward: pre = 1/deg' (avg), post = 1        (gather over the TRANSPOSED CSR)
//   learn_eps=True : selfA = 0, selfB = 1 + eps[layer], deg' = deg
//   learn_eps=False: selfA = 1, selfB = 0,             deg' = deg + 1  (self loops,
//                    graphcnn.py:97-102)
// In backward it also produces d eps[layer] = sum dpooled * h (fp64 partials).
#include "gnm_common.h"
#include <stdlib.h>
#include <string.h>

#include "gnm_agg_args.h"

template <int S>
__device__ __forceinline__ unsigned row_bcast16(unsigned v) {
    // DPP row_newbcast:S -- every lane of a 16-lane row reads lane S of its row
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + S, 0xf, 0xf, false);
}

// x' + y' after v_permlane32_swap: lanes 0-31 get x[l] + x[l+32], lanes 32-63 get y[l-32] + y[l]
__device__ __forceinline__ float swap_add32_1(float x, float y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float4 swap_add32(const float4 x, const float4 y) {
    return make_float4(swap_add32_1(x.x, y.x), swap_add32_1(x.y, y.y), swap_add32_1(x.z, y.z), swap_add32_1(x.w, y.w));
}
// after v_permlane16_swap: 16-lane rows 0,2 get x[row]+x[row+1], rows 1,3 get y[row-1]+y[row]
__device__ __forceinline__ float swap_add16_1(float x, float y) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float4 swap_add16(const float4 x, const float4 y) {
    return make_float4(swap_add16_1(x.x, y.x), swap_add16_1(x.y, y.y), swap_add16_1(x.z, y.z), swap_add16_1(x.w, y.w));
}

// x[l ^ 8] within each 16-lane DPP row (row_ror:8)
__device__ __forceinline__ float ror8(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));
}
// x[l ^ 2] / x[l ^ 1] within each quad (DPP quad_perm [2,3,0,1] / [1,0,3,2])
__device__ __forceinline__ float qxor2(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));
}
__device__ __forceinline__ float qxor1(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));
}
// lane-indexed read of a lane vector (per-lane index i < 16): v[i] -- two 16-lane-row broadcasts would need a uniform
// index, so this is a ds_bpermute; used once per 8-row group (degree for "average" only)
__device__ __forceinline__ int bnd_of(int v, int i) { return __shfl(v, i, 64); }

template <int S>
__device__ __forceinline__ unsigned bcast8(unsigned v) {
    // 8-lane groups: lanes 0-7 of each 16-lane DPP row read lane S of the row, lanes 8-15 read lane 8 + S
    int r = __builtin_amdgcn_update_dpp(0, (int)v, 0x150 + S, 0xf, 0x3, false);
    r = __builtin_amdgcn_update_dpp(r, (int)v, 0x150 + 8 + S, 0xf, 0xc, false);
    return (unsigned)r;
}

__device__ __forceinline__ void acc4(float4& a, const float4 v) {
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
}
__device__ __forceinline__ void acc4(float4& a, const f32x4 v) {
    a.x += v[0]; a.y += v[1]; a.z += v[2]; a.w += v[3];
}

// LDS is addressed through 32-bit address-space-3 pointers built from integers, so
// that "row base (DPP broadcast) + lane chunk + LDS base" is ONE v_add_u32_dpp
// feeding ds_read_b128 (a generic char* + offset costs an extra VALU per gather).
typedef __attribute__((address_space(3))) const f32x4* lds_cf4p;
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {
    return *(lds_cf4p)(uintptr_t)addr;
}

typedef unsigned int agg_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const agg_u32x2* lds_cu2p;
__device__ __forceinline__ agg_u32x2 lds_read_u32x2(unsigned addr) { return *(lds_cu2p)(uintptr_t)addr; }
typedef __attribute__((address_space(3))) unsigned short* lds_u16p;
__device__ __forceinline__ unsigned lds_read_u16(unsigned addr) { return *(lds_u16p)(uintptr_t)addr; }
__device__ __forceinline__ void lds_write_u16(unsigned addr, unsigned v) { *(lds_u16p)(uintptr_t)addr = (unsigned short)v; }

#define GNM_STEP16(S) acc4(acc, lds_read16(row_bcast16<S>(valb) + subb));

// Column id at (uniform base, per-lane 32-bit index): written as base + zero-extended BYTE offset so the load
// takes the SGPR-base + 32-bit VGPR-offset form (one v_lshl_add_u32 per load instead of a 64-bit address built
// from three VALU instructions).  A graph's block holds < 2^30 ids.
__device__ __forceinline__ unsigned load_id(const uint16_t* base, unsigned byte_off) {
    return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(base) + byte_off);
}

// MODE (32-float slices, LPR == 8, only; round 4 -- what gnm_agg16_kernel does for the 64-wide shape):
//   1 = forward PROLOGUE: x is the Z of the layer below's last Linear; the tile load applies that layer's outer
//       BatchNorm + ReLU (p_scale / p_shift), writes the activation (p_hout, optional) and the graph readout (p_gf):
//       gnm_bn_relu_readout's pass (45 us of 128-wide rows at configs[3]) rides on loads this kernel issues anyway;
//   2 = backward STATS: the epilogue adds the readout / discriminator terms, applies the ReLU mask of the layer
//       below and reduces that layer's BatchNorm-backward column sums (s_partial), and takes d eps from the rows it
//       holds (h recomputed from sZ): gnm_bn_relu_bwd_stats's pass (75 us) and the hfwd read of phase A go away.
template <int LPR, int MODE = 0>
__global__ void __launch_bounds__(1024) gnm_agg_kernel(const AggArgs p) {
    constexpr int FS = LPR * 4;        // floats per LDS row
    constexpr int SLOTS = 64 / LPR;    // neighbour rows read per wave-instruction
    constexpr bool PRO = MODE == 1, STATS = MODE == 2;
    static_assert(MODE == 0 || LPR == 8, "fused forms: 32-float slices only");
#ifdef GNM_AGG16_TUNING       // in-kernel timeline (tools/agg_timeline.py)
#define GNM_GSTAMP(k)                                                                                         \
    if (p.stamps && (threadIdx.x & 63) == 0)                                                                  \
        p.stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define GNM_GSTAMP(k)
#endif
    GNM_GSTAMP(0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* tile = reinterpret_cast<float4*>(smem);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int b = blockIdx.x / p.nslices;
    const int sl = blockIdx.x - b * p.nslices;
    const int row0 = p.node_off[b];
    const int n = p.node_off[b + 1] - row0;
    const int col0 = sl * FS;
    const int32_t* rp = p.rowptr + p.b_rp_off[b];
    const uint16_t* cl = p.col + p.b_col_off[b];
    const int32_t* drp = p.deg_rowptr + p.b_deg_off[b];
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const bool vec_in = ((p.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
    const bool vec_h = p.hfwd && ((p.ldh & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.hfwd) & 15) == 0);
    const bool prescale = p.backward && p.average;

    // ---- phase A: HBM -> LDS, coalesced; optional 1/deg pre-scale and d-eps dot ----
    double dot = 0.0;
    int i0 = tid;
    // d eps: dot(x, hfwd) on the way in, or -- STATS launches without hfwd -- from the rows the epilogue holds.  The
    // launcher only starts the fused forms on full-width, 16-byte addressable slices, and a thread keeps the column
    // chunk tid & (LPR - 1) for all its rows (the workgroup size is a multiple of LPR).
    const bool dot_a = p.deps_partial && (!STATS || p.hfwd);
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (PRO) {
        psc = *reinterpret_cast<const float4*>(p.p_scale + col0 + 4 * (tid & (LPR - 1)));
        psh = *reinterpret_cast<const float4*>(p.p_shift + col0 + 4 * (tid & (LPR - 1)));
    }
    auto prologue = [&](int i, float4 w) -> float4 {      // (PRO) the activation of tile element i, written and summed
        w.x = gnm_relu(w.x * psc.x + psh.x); w.y = gnm_relu(w.y * psc.y + psh.y);
        w.z = gnm_relu(w.z * psc.z + psh.z); w.w = gnm_relu(w.w * psc.w + psh.w);
        if (p.p_hout) *reinterpret_cast<float4*>(p.p_hout + (size_t)(row0 + i / LPR) * p.p_ldh + col0 + 4 * (i & (LPR - 1))) = w;
        csum.x += w.x; csum.y += w.y; csum.z += w.z; csum.w += w.w;
        return w;
    };
    // full-width, 16-byte addressable slices: 8 x 16 B per thread in flight (the loop below issues ONE load per trip,
    // inside a lane-dependent branch, i.e. one memory round trip per 16 KB of the tile: 16-32 us of a 128 KB tile)
    if (vec_in && col0 + FS <= p.F && (!dot_a || vec_h)) {       // wave-uniform
        constexpr int UA = 8;
        const int total = n * LPR;
        for (; i0 + (UA - 1) * nthreads < total; i0 += UA * nthreads) {
            float4 v[UA];
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int i = i0 + u * nthreads;
                const int r = i / LPR, c = i - r * LPR;
                v[u] = *reinterpret_cast<const float4*>(p.x + (size_t)(row0 + r) * p.ldx + col0 + 4 * c);
            }
            if (dot_a) {
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    const int i = i0 + u * nthreads;
                    const int r = i / LPR, c = i - r * LPR;
                    const float4 h = *reinterpret_cast<const float4*>(p.hfwd + (size_t)(row0 + r) * p.ldh + col0 + 4 * c);
                    dot += (double)v[u].x * h.x + (double)v[u].y * h.y + (double)v[u].z * h.z + (double)v[u].w * h.w;
                }
            }
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int i = i0 + u * nthreads;
                float4 w = v[u];
                if (prescale) {
                    const int r = i / LPR;
                    const float d = (float)(drp[r + 1] - drp[r] + p.self_loop);
                    w.x /= d; w.y /= d; w.z /= d; w.w /= d;
                }
                if constexpr (PRO) w = prologue(i, w);
                tile[i] = w;
            }
        }
    }
    for (int i = i0; i < n * LPR; i += nthreads) {
        const int r = i / LPR;
        const int c = i - r * LPR;
        const int cc = col0 + 4 * c;
        const float* src = p.x + (size_t)(row0 + r) * p.ldx + cc;
        float4 v;
        if (vec_in && cc + 3 < p.F) {
            v = *reinterpret_cast<const float4*>(src);
        } else {
            v.x = (cc + 0 < p.F) ? src[0] : 0.f;
            v.y = (cc + 1 < p.F) ? src[1] : 0.f;
            v.z = (cc + 2 < p.F) ? src[2] : 0.f;
            v.w = (cc + 3 < p.F) ? src[3] : 0.f;
        }
        if (dot_a) {
            const float* hs = p.hfwd + (size_t)(row0 + r) * p.ldh + cc;
            float4 h;
            if (vec_h && cc + 3 < p.F) {
                h = *reinterpret_cast<const float4*>(hs);
            } else {
                h.x = (cc + 0 < p.F) ? hs[0] : 0.f;
                h.y = (cc + 1 < p.F) ? hs[1] : 0.f;
                h.z = (cc + 2 < p.F) ? hs[2] : 0.f;
                h.w = (cc + 3 < p.F) ? hs[3] : 0.f;
            }
            dot += (double)v.x * h.x + (double)v.y * h.y + (double)v.z * h.z + (double)v.w * h.w;
        }
        if (prescale) {
            const float d = (float)(drp[r + 1] - drp[r] + p.self_loop);
            v.x /= d; v.y /= d; v.z /= d; v.w /= d;
        }
        if constexpr (PRO) v = prologue(i, v);
        tile[i] = v;
    }
    constexpr int ZR = LPR == 8 ? 2 : 1;       // zero rows behind the tile (LPR == 8: one of each parity, see phase B)
    if (tid < ZR * LPR) tile[n * LPR + tid] = make_float4(0.f, 0.f, 0.f, 0.f);  // row n (, n + 1) = zeros (padding slots)
    // row offsets staged behind the tile (wave-cooperative path only: the narrow path keeps its id block there)
    int* rp_s = reinterpret_cast<int*>(smem + (size_t)(n + ZR) * (FS * 4));
    const bool stage_rp = !(LPR <= 4 && p.ids_in_lds);
    if (stage_rp)
        for (int i = tid; i <= n; i += nthreads) rp_s[i] = rp[i];
    if (LPR == 8 && tid == 0) rp_s[n + 1] = nthreads >> 6;      // the group ticket (phase B): every wave's first group is its own number
    // LDS behind the row offsets (LPR == 8): the waves' id scratch (16 x 1 KB, ids_in_lds == 2), then the prologue's
    // readout shares ([waves][8] float4)
    const size_t scr_off = ((size_t)(n + ZR) * (FS * 4) + (size_t)(n + 2) * 4 + 16 + 15) & ~(size_t)15;
    float4* const rsum = reinterpret_cast<float4*>(smem + scr_off + (p.ids_in_lds == 2 ? 16 * 1024 : 0));
    if constexpr (PRO) {
        if (p.p_gf) {      // the 8 lanes of a wave that share a column chunk (lane bits 3-5), then the waves (below)
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                csum.x += __shfl_xor(csum.x, off, 64); csum.y += __shfl_xor(csum.y, off, 64);
                csum.z += __shfl_xor(csum.z, off, 64); csum.w += __shfl_xor(csum.w, off, 64);
            }
            if ((tid & 63) < 8) rsum[(tid >> 6) * 8 + (tid & 7)] = csum;
        }
    }
    GNM_GSTAMP(1)
    GNM_GSTAMP(2)
    __syncthreads();
    GNM_GSTAMP(3)
    if constexpr (PRO) {
        if (p.p_gf && tid < 8) {      // graph readout of the activation just formed (graphcnn.py:229), in wave order
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int w = 0; w < (nthreads >> 6); ++w) acc4(t, rsum[w * 8 + tid]);
            if (p.p_gf_avg) {
                const float inv = 1.f / (float)n;
                t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
            }
            *reinterpret_cast<float4*>(p.p_gf + (size_t)b * p.p_ldgf + col0 + 4 * tid) = t;
        }
    }

    // ---- phase B, narrow features (FS <= 16 floats): one THREAD per (row, 16-B chunk) ------
    // The rows are so short that the wave-cooperative scheme below spends its time in
    // cross-lane reductions and dependent loads.  Here the graph's column ids are staged in
    // LDS as well (p.ids_in_lds, decided by the launcher), a thread walks its row's ids 8 at a
    // time (8 independent LDS id reads, then 8 row reads) and owns its output chunk outright.
    if constexpr (LPR <= 4) {
        if (p.ids_in_lds) {
            // ids[e] = cl[e], staged with 16-B copies, 4 per thread in flight (64 KB per workgroup): with 4-B
            // copies only 16 KB were in flight per CU and this copy alone bounded the kernel at ~2 TB/s.
            // The LDS image is shifted by the source's misalignment so that both sides are 16-B aligned
            // together; the copy may run up to 7 ids before / 8 ids past the graph's block (the arena keeps
            // readable slack after the last block, and a block never starts at the buffer's first 16 bytes
            // unless it is aligned).
            const int mis = (int)((reinterpret_cast<uintptr_t>(cl) & 15) >> 1);          // ids before alignment
            uint16_t* ids = reinterpret_cast<uint16_t*>(smem + (((size_t)(n + 1) * (FS * 4) + 15) & ~(size_t)15)) + mis;
            const int nnz = p.y ? rp[n] : 0;            // y == null (d-eps only): nothing to gather
            {
                const uint4* src = reinterpret_cast<const uint4*>(cl - mis);
                uint4* dst = reinterpret_cast<uint4*>(ids - mis);
                const int nq = (nnz + mis + 7) >> 3;
                int e = tid;
                for (; e + 3 * nthreads < nq; e += 4 * nthreads) {
                    const uint4 v0 = src[e], v1 = src[e + nthreads], v2 = src[e + 2 * nthreads], v3 = src[e + 3 * nthreads];
                    dst[e] = v0; dst[e + nthreads] = v1; dst[e + 2 * nthreads] = v2; dst[e + 3 * nthreads] = v3;
                }
                for (; e < nq; e += nthreads) dst[e] = src[e];
            }
            __syncthreads();
            const float selfB2 = p.self_loop ? 0.f : (p.eps ? 1.f + *p.eps : 1.f);
            const bool vec_out2 = ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0);
            for (int w = tid; p.y && w < n * LPR; w += nthreads) {
                const int v = w / LPR, sub2 = w - v * LPR;
                const int beg = rp[v], end = rp[v + 1];
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                int e = beg;
                for (; e + 8 <= end; e += 8) {
                    unsigned id[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) id[j] = ids[e + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc4(acc, tile[id[j] * LPR + sub2]);
                }
                for (; e < end; ++e) acc4(acc, tile[(unsigned)ids[e] * LPR + sub2]);
                const float4 self = tile[v * LPR + sub2];
                if (p.self_loop) acc4(acc, self);
                if (!p.backward && p.average) {
                    const float d = (float)(end - beg + p.self_loop);
                    acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d;
                }
                const int cc = col0 + 4 * sub2;
                if (!p.self_loop) {
                    float4 sb = self;
                    if (prescale) {
                        const float* src = p.x + (size_t)(row0 + v) * p.ldx + cc;
                        sb.x = (cc + 0 < p.F) ? src[0] : 0.f;
                        sb.y = (cc + 1 < p.F) ? src[1] : 0.f;
                        sb.z = (cc + 2 < p.F) ? src[2] : 0.f;
                        sb.w = (cc + 3 < p.F) ? src[3] : 0.f;
                    }
                    acc.x += selfB2 * sb.x; acc.y += selfB2 * sb.y; acc.z += selfB2 * sb.z; acc.w += selfB2 * sb.w;
                }
                float* dst = p.y + (size_t)(row0 + v) * p.ldy + cc;
                if (vec_out2 && cc + 3 < p.F) {
                    *reinterpret_cast<float4*>(dst) = acc;
                } else {
                    if (cc + 0 < p.F) dst[0] = acc.x;
                    if (cc + 1 < p.F) dst[1] = acc.y;
                    if (cc + 2 < p.F) dst[2] = acc.z;
                    if (cc + 3 < p.F) dst[3] = acc.w;
                }
            }
            if (p.deps_partial) {
                __syncthreads();
                double* red = reinterpret_cast<double*>(smem);
                const double ws = wave_sum_d(dot);
                if ((tid & 63) == 0) red[tid >> 6] = ws;
                __syncthreads();
                if (tid == 0) {
                    double s2 = 0.0;
                    for (int k = 0; k < (nthreads >> 6); ++k) s2 += red[k];
                    p.deps_partial[blockIdx.x] = s2;
                }
            }
            return;
        }
    }

    // ---- phase B: gather neighbour rows from LDS --------------------------------
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = nthreads >> 6;
    const int sub = lane & (LPR - 1);          // 16-B chunk of the row this lane owns
    const int slot = lane / LPR;               // which of the SLOTS concurrent neighbours
    const unsigned subb = lds_base + (unsigned)sub * 16u;   // LDS byte address of this lane's chunk in row 0
    const int jlane = sub * SLOTS + slot;      // edge (within a 64-chunk) whose id this lane fetches
    const unsigned zero_row_b = (unsigned)n * (FS * 4);
    const float selfB = p.self_loop ? 0.f : (p.eps ? 1.f + *p.eps : 1.f);
    const bool vec_out = ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0);

    // ---- 32-float slices (8 lanes per row, 8 neighbours per wave-instruction): GROUPS OF 8 ROWS per wave ----------
    // Row r of a group is gathered by all 8 lane groups (each takes every 8th neighbour); the 8 x 8 partial sums are
    // then combined by a transposing butterfly -- v_permlane32_swap, v_permlane16_swap, DPP row_ror:8 -- after which lane
    // group k owns the whole sum of row k and all 64 lanes run the epilogue and store 1 KiB.  No ds_bpermute and no
    // per-row global round trip: the 9 row offsets come from LDS as a lane vector fetched one group ahead, the 8 id
    // chunks of a group are requested together.
    if constexpr (LPR == 8) {
        // Bank conflicts (round 4).  A 128-byte row lies on banks 0-31 or 32-63 by the parity of its index, and the
        // hardware serves a ds_read_b128 in 16-lane groups that put the neighbours of slots (0, 3), (1, 2), (4, 7),
        // (5, 6) on the same cycle: equal parity there = a 2-way conflict (a third of this kernel's LDS cycles in
        // round 3's PMC).  The arena orders every CSR row so that slots 0, 1, 4, 5 get even ids and slots 2, 3, 6, 7
        // odd ones while the row has both (host.cpp gnm_csr_parity_order); padding slots read the zero row of the
        // parity their slot wants (rows n and n + 1 are both zero).
        const unsigned zero_row_pad = (unsigned)(n + ((n ^ (slot >> 1)) & 1)) * (FS * 4);
        const int ngroups = p.y ? (n + 7) >> 3 : 0;
        const unsigned jl2 = 2u * (unsigned)jlane;
        const bool hi8 = (lane & 8) != 0;
        // Round 4: groups are handed out by an LDS ticket (a wave's first group is its own number, the counter starts
        // at nwaves) -- the static deal left the median wave idle for 17 % of the workgroup's life behind the waves
        // whose rows happened to have more than 32 neighbours (in-kernel timeline, profiles/r04_c4_timeline.md) -- and
        // the NEXT group's row bounds and first id chunks are requested in the middle of the current one (one exposed
        // L2 round trip per group before).  A row's own sum does not depend on which wave takes it.
        int* const ticket = rp_s + n + 1;
        // wave-private id scratch behind the row offsets (8 rows x 64 positions x 2 bytes per wave), when the launcher
        // found room for it (ids_in_lds == 2)
        const bool stage8 = p.ids_in_lds == 2;
        const unsigned scr0 = lds_base + (unsigned)scr_off + (unsigned)wave * 1024u;
        // scratch row layout [slot][step]: the 8 ids a slot reads over the steps of a row are 16 contiguous bytes, so
        // steps 0-3 (4-7) come with ONE ds_read_b64 each -- eight 2-byte reads per row cost half as many LDS cycles
        // as the row gather itself (the kernel became LDS-pipe bound once the DPP moves were gone)
        const unsigned scr_w = scr0 + 2u * (unsigned)((jlane & 7) * 8 + (jlane >> 3)), scr_r = scr0 + 16u * (unsigned)slot;
        const unsigned zero_id = (unsigned)(n + ((n ^ (slot >> 1)) & 1));
        int g = wave;
        int rpv = 0;
        unsigned raw[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) raw[r] = 0u;
        if (g < ngroups) {
            rpv = rp_s[min(8 * g + (lane & 15), n)];
#pragma unroll
            for (int r = 0; r < 8; ++r) raw[r] = load_id(cl, 2u * (unsigned)__builtin_amdgcn_readlane(rpv, r) + jl2);
        }
        // STATS: per-lane column sums of the rows this lane group ends up owning, and the per-column constants of the
        // layer below (chunk `sub` of this slice).  The deal of groups is STATIC there (wave, wave + W, ...): the sums
        // are accumulated per wave, and with tickets their order -- hence their last bits -- would change from launch
        // to launch (every reduction of this library has a fixed order).
        float4 ss1 = make_float4(0.f, 0.f, 0.f, 0.f), ss2 = ss1, s_pb = ss1, s_ub = ss1;
        float4 lsc = ss1, lsh = ss1, lmu = ss1;
        if constexpr (STATS) {
            lsc = *reinterpret_cast<const float4*>(p.s_scale + col0 + 4 * sub);
            lsh = *reinterpret_cast<const float4*>(p.s_shift + col0 + 4 * sub);
            lmu = *reinterpret_cast<const float4*>(p.s_mean + col0 + 4 * sub);    // (rstd multiplies the column sums at the end)
            if (p.s_dpool) {
                s_pb = *reinterpret_cast<const float4*>(p.s_dpool + (size_t)b * p.ld_dpool + col0 + 4 * sub);
                if (p.s_avg) {
                    const float w = 1.0f / (float)n;
                    s_pb.x *= w; s_pb.y *= w; s_pb.z *= w; s_pb.w *= w;
                }
            }
            if (p.s_dsc1) s_ub = *reinterpret_cast<const float4*>(p.s_U + (size_t)b * p.ld_U + col0 + 4 * sub);
        }
        while (g < ngroups) {                                       // wave-uniform
            int gn = g + nwaves;
            if constexpr (!STATS) {
                gn = 0;
                if (lane == 0) gn = atomicAdd(ticket, 1);            // (read after the first four rows)
            }
            int bnd[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) bnd[i] = __builtin_amdgcn_readlane(rpv, i);
            const int v = 8 * g + slot;
            const float4 self = tile[min(v, n) * LPR + sub];       // ahead of the gather (row n = zeros)
            float4 zrow = make_float4(0.f, 0.f, 0.f, 0.f);
            float dsc_v = 0.f;
            if constexpr (STATS) {     // requested ahead of the gather, consumed in the epilogue
                const int vc = row0 + min(v, n - 1);
                zrow = *reinterpret_cast<const float4*>(p.sZ + (size_t)vc * p.ldsz + col0 + 4 * sub);
                if (p.s_dsc1) dsc_v = p.s_dsc1[vc];
            }
            float4 acc[8];
            int nrpv = 0;
            agg_u32x2 idn = {0u, 0u};
            if (stage8) {
                // ids through a wave-private LDS scratch (round 4): a lane's id of row r goes to scratch[r][position]
                // (padding positions: the zero row of the slot's parity), and step S of slot k reads position 8 S + k
                // back with an immediate offset -- one 2-byte LDS read and one shift-add per step instead of two
                // bank-masked DPP moves, a zero fill and an add (the kernel is VALU-issue bound: ~35 VALU per row).
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int cnt = min(64, bnd[r + 1] - bnd[r]);
                    lds_write_u16(scr_w + 128u * r, (jlane < cnt) ? raw[r] : zero_id);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                idn = lds_read_u32x2(scr_r);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r == 4) {                                       // the next group's bounds: an LDS read under rows 4-7
                    gn = __builtin_amdgcn_readfirstlane(gn);
                    nrpv = rp_s[min(8 * gn + (lane & 15), n)];      // (exhausted ticket: every bound = rp[n], reads the slack)
                }
                const int beg = bnd[r], end = bnd[r + 1];
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                int e0 = beg;
                if (stage8) {
                    // (every position of the scratch row holds a valid id -- padding = a zero row -- so the first four
                    //  steps need no guard, and the ids of row r + 1 are requested before row r's rows are waited for:
                    //  one exposed LDS round trip per row instead of two)
                    const unsigned rb = scr_r + 128u * r;
                    const unsigned i0 = idn.x & 0xFFFFu, i1 = idn.x >> 16, i2 = idn.y & 0xFFFFu, i3 = idn.y >> 16;
                    if (r < 7) idn = lds_read_u32x2(rb + 128u);
                    {
                        const f32x4 t0 = lds_read16((i0 << 7) + subb), t1 = lds_read16((i1 << 7) + subb),
                                    t2 = lds_read16((i2 << 7) + subb), t3 = lds_read16((i3 << 7) + subb);
                        acc4(a, (t0 + t1) + (t2 + t3));
                    }
                    if (end - beg > 32) {                          // wave-uniform
                        const agg_u32x2 jj = lds_read_u32x2(rb + 8u);
                        const unsigned j0 = jj.x & 0xFFFFu, j1 = jj.x >> 16, j2 = jj.y & 0xFFFFu, j3 = jj.y >> 16;
                        const f32x4 t0 = lds_read16((j0 << 7) + subb), t1 = lds_read16((j1 << 7) + subb),
                                    t2 = lds_read16((j2 << 7) + subb), t3 = lds_read16((j3 << 7) + subb);
                        acc4(a, (t0 + t1) + (t2 + t3));
                    }
                    e0 = beg + 64;                                  // (rows of more than 64 neighbours: the chunks below)
                }
                for (; e0 < end; e0 += 64) {
                    const int cnt = min(64, end - e0);
                    const unsigned rw = (e0 == beg) ? raw[r] : load_id(cl, 2u * (unsigned)e0 + jl2);
                    const unsigned valb = (jlane < cnt) ? rw * (FS * 4) : zero_row_pad;
                    {
                        const f32x4 t0 = lds_read16(bcast8<0>(valb) + subb), t1 = lds_read16(bcast8<1>(valb) + subb),
                                    t2 = lds_read16(bcast8<2>(valb) + subb), t3 = lds_read16(bcast8<3>(valb) + subb);
                        acc4(a, (t0 + t1) + (t2 + t3));
                    }
                    if (cnt > 32) {                                // wave-uniform: steps 4..7 hold something
                        const f32x4 t0 = lds_read16(bcast8<4>(valb) + subb), t1 = lds_read16(bcast8<5>(valb) + subb),
                                    t2 = lds_read16(bcast8<6>(valb) + subb), t3 = lds_read16(bcast8<7>(valb) + subb);
                        acc4(a, (t0 + t1) + (t2 + t3));
                    }
                }
                acc[r] = a;
            }
            // the next group's first id chunks: in flight under the combine, the epilogue and the stores
#pragma unroll
            for (int r = 0; r < 8; ++r) raw[r] = load_id(cl, 2u * (unsigned)__builtin_amdgcn_readlane(nrpv, r) + jl2);
            // transposing combine: lane bit 5 picks rows {0-3 | 4-7}, bit 4 {r | r+2}, bit 3 {r | r+1}
            const float4 b0 = swap_add32(acc[0], acc[4]), b1 = swap_add32(acc[1], acc[5]);
            const float4 b2 = swap_add32(acc[2], acc[6]), b3 = swap_add32(acc[3], acc[7]);
            const float4 c0 = swap_add16(b0, b2), c1 = swap_add16(b1, b3);
            const float4 mine = hi8 ? c1 : c0, theirs = hi8 ? c0 : c1;
            float4 tot;
            tot.x = mine.x + ror8(theirs.x); tot.y = mine.y + ror8(theirs.y);
            tot.z = mine.z + ror8(theirs.z); tot.w = mine.w + ror8(theirs.w);
            if (v < n) {
                const int beg = bnd_of(rpv, slot), end = bnd_of(rpv, slot + 1);
                if (p.self_loop) acc4(tot, self);
                if (!p.backward && p.average) {
                    const float d = (float)(end - beg + p.self_loop);   // 0/0 -> NaN as in the reference
                    tot.x /= d; tot.y /= d; tot.z /= d; tot.w /= d;
                }
                const int cc = col0 + 4 * sub;
                if (!p.self_loop) {
                    float4 sb = self;
                    if (prescale) {   // the tile holds dp/deg; the (1+eps) term needs dp itself
                        const float* src = p.x + (size_t)(row0 + v) * p.ldx + cc;
                        sb.x = (cc + 0 < p.F) ? src[0] : 0.f;
                        sb.y = (cc + 1 < p.F) ? src[1] : 0.f;
                        sb.z = (cc + 2 < p.F) ? src[2] : 0.f;
                        sb.w = (cc + 3 < p.F) ? src[3] : 0.f;
                    }
                    tot.x += selfB * sb.x; tot.y += selfB * sb.y; tot.z += selfB * sb.z; tot.w += selfB * sb.w;
                    if constexpr (STATS) {
                        if (p.deps_partial && !p.hfwd) {
                            // d eps += dpooled[v] . h[v], h = relu(bn_lo(Z[v])) recomputed as the forward formed it
                            const float hx = gnm_relu(zrow.x * lsc.x + lsh.x), hy = gnm_relu(zrow.y * lsc.y + lsh.y);
                            const float hz = gnm_relu(zrow.z * lsc.z + lsh.z), hw = gnm_relu(zrow.w * lsc.w + lsh.w);
                            dot += (double)(sb.x * hx + sb.y * hy) + (double)(sb.z * hz + sb.w * hw);
                        }
                    }
                }
                if constexpr (STATS) {
                    // total gradient at this layer output = aggregation backward + readout + discriminator terms
                    tot.x += s_pb.x + dsc_v * s_ub.x; tot.y += s_pb.y + dsc_v * s_ub.y;
                    tot.z += s_pb.z + dsc_v * s_ub.z; tot.w += s_pb.w + dsc_v * s_ub.w;
                    if (p.s_dsc1 && row0 + v < p.n_batch) {     // rows perm[g] < B of the shuffled branch (graphcnn.py:242)
                        const int gq = gnm_perm_entry(p.s_inv_perm[row0 + v], p.n_batch);
                        const float s2 = p.s_s2sum[gq];
                        const float4 uq = *reinterpret_cast<const float4*>(p.s_U + (size_t)gq * p.ld_U + col0 + 4 * sub);
                        tot.x += s2 * uq.x; tot.y += s2 * uq.y; tot.z += s2 * uq.z; tot.w += s2 * uq.w;
                    }
                    if (!(zrow.x * lsc.x + lsh.x > 0.f)) tot.x = 0.f;
                    if (!(zrow.y * lsc.y + lsh.y > 0.f)) tot.y = 0.f;
                    if (!(zrow.z * lsc.z + lsh.z > 0.f)) tot.z = 0.f;
                    if (!(zrow.w * lsc.w + lsh.w > 0.f)) tot.w = 0.f;
                    ss1.x += tot.x; ss1.y += tot.y; ss1.z += tot.z; ss1.w += tot.w;
                    ss2.x += tot.x * (zrow.x - lmu.x); ss2.y += tot.y * (zrow.y - lmu.y);
                    ss2.z += tot.z * (zrow.z - lmu.z); ss2.w += tot.w * (zrow.w - lmu.w);
                }
                float* dst = p.y + (size_t)(row0 + v) * p.ldy + cc;
                if (vec_out && col0 + FS <= p.F) {          // wave-uniform: ONE 16-byte store per lane (see agg16)
                    const f32x4 t4 = {tot.x, tot.y, tot.z, tot.w};
                    __builtin_nontemporal_store(t4, reinterpret_cast<f32x4*>(dst));
                } else {
                    if (cc + 0 < p.F) dst[0] = tot.x;
                    if (cc + 1 < p.F) dst[1] = tot.y;
                    if (cc + 2 < p.F) dst[2] = tot.z;
                    if (cc + 3 < p.F) dst[3] = tot.w;
                }
            }
            rpv = nrpv;
            g = gn;
        }
        GNM_GSTAMP(63)
        if constexpr (STATS) {     // column sums: the 8 lane groups of a wave, then the waves, in a fixed order
            __syncthreads();        // (everyone is done reading the tile: its first bytes are reused)
            double* sred = reinterpret_cast<double*>(smem) + 64;      // [nwaves][2][32] (after the d-eps slots)
            const float v1[4] = {ss1.x, ss1.y, ss1.z, ss1.w}, v2[4] = {ss2.x, ss2.y, ss2.z, ss2.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double d1 = (double)v1[c], d2 = (double)v2[c];
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) {
                    d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64);
                }
                if (lane < 8) {
                    sred[(wave * 2 + 0) * 32 + 4 * sub + c] = d1;
                    sred[(wave * 2 + 1) * 32 + 4 * sub + c] = d2;
                }
            }
            __syncthreads();
            if (tid < 64) {
                const int which = tid >> 5, col = tid & 31;
                double sum = 0.0;
                for (int w = 0; w < nwaves; ++w) sum += sred[(w * 2 + which) * 32 + col];
                if (which) sum *= (double)p.s_rstd[col0 + col];      // sum G (Z - mean) -> sum G xhat
                p.s_partial[((size_t)b * 2 + which) * p.F + col0 + col] = sum;
            }
        }
    } else if constexpr (LPR == 2) {
        // ---- 8-float rows (the input layer, F0 <= 8): 2 lanes per neighbour row, 32 neighbours per wave-instruction,
        // GROUPS OF 16 ROWS per wave.  Lane bit 2 is the 16-B chunk; the other five bits number the 32 neighbour slots,
        // so both lanes of a slot load the same column id themselves (no cross-lane traffic at all), and the 16 x 32
        // partial sums are transposed and added over lane bits 5, 4, 3 (permlane swaps, DPP row_ror:8) and 1, 0
        // (DPP quad_perm): lanes {bit5, bit4, bit3, bit1} = k end up with row k.
        const int ngroups = p.y ? (n + 15) >> 4 : 0;
        const int q2 = (lane >> 2) & 1;                                   // chunk of the 32-B row
        const unsigned p5 = (unsigned)(((lane >> 3) << 2) | (lane & 3));   // neighbour slot 0..31
        const unsigned sub2b = lds_base + (unsigned)q2 * 16u;
        const unsigned zero2 = sub2b + (unsigned)n * 32u;
        const bool b3 = (lane & 8) != 0, b1 = (lane & 2) != 0;
        const int krow = ((lane >> 3) << 1) | ((lane >> 1) & 1);          // row of the group this lane ends up owning
        int g = wave;
        int rpv = 0, nrpv = 0;
        if (g < ngroups) rpv = rp_s[min(16 * g + (lane & 31), n)];
        unsigned nr0 = 0, nr1 = 0, nr2 = 0, nr3 = 0;                        // ids of the NEXT row (4 chunks of 32)
        if (g < ngroups) {
            const unsigned o = 2u * ((unsigned)__builtin_amdgcn_readlane(rpv, 0) + p5);
            nr0 = load_id(cl, o); nr1 = load_id(cl, o + 64u); nr2 = load_id(cl, o + 128u); nr3 = load_id(cl, o + 192u);
        }
        for (; g < ngroups; g += nwaves) {
            const bool more = g + nwaves < ngroups;
            if (more) nrpv = rp_s[min(16 * (g + nwaves) + (lane & 31), n)];
            const int v = 16 * g + krow;
            const float4 self = tile[min(v, n) * 2 + q2];
            float4 acc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int beg = __builtin_amdgcn_readlane(rpv, r), end = __builtin_amdgcn_readlane(rpv, r + 1);
                const unsigned r0 = nr0, r1 = nr1, r2 = nr2, r3 = nr3;
                // the next row's ids (row r + 1 of this group, or row 0 of this wave's next group), unconditionally
                {
                    const int nb = (r < 15) ? end : __builtin_amdgcn_readlane(nrpv, 0);
                    if (r < 15 || more) {                              // wave-uniform
                        const unsigned o = 2u * ((unsigned)nb + p5);
                        nr0 = load_id(cl, o); nr1 = load_id(cl, o + 64u); nr2 = load_id(cl, o + 128u); nr3 = load_id(cl, o + 192u);
                    }
                }
                const int cnt = end - beg;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cnt > 0) {
                    const unsigned a0 = ((int)p5 < cnt) ? (r0 << 5) + sub2b : zero2;
                    const unsigned a1 = ((int)p5 + 32 < cnt) ? (r1 << 5) + sub2b : zero2;
                    if (cnt > 64) {
                        const unsigned a2 = ((int)p5 + 64 < cnt) ? (r2 << 5) + sub2b : zero2;
                        const unsigned a3 = ((int)p5 + 96 < cnt) ? (r3 << 5) + sub2b : zero2;
                        const f32x4 t0 = lds_read16(a0), t1 = lds_read16(a1), t2 = lds_read16(a2), t3 = lds_read16(a3);
                        acc4(a, (t0 + t1) + (t2 + t3));
                        for (int e0 = beg + 128; e0 < end; e0 += 32) {       // degree > 128: fetched in place
                            const unsigned rw = load_id(cl, 2u * ((unsigned)e0 + p5));
                            const unsigned ax = ((int)p5 < end - e0) ? (rw << 5) + sub2b : zero2;
                            acc4(a, lds_read16(ax));
                        }
                    } else {
                        const f32x4 t0 = lds_read16(a0), t1 = lds_read16(a1);
                        acc4(a, t0 + t1);
                    }
                }
                acc[r] = a;
            }
            // transposing combine over lane bits 5, 4, 3, 1, then the plain sum over bit 0
            float4 b8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b8[j] = swap_add32(acc[j], acc[j + 8]);
            float4 c4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c4[j] = swap_add16(b8[j], b8[j + 4]);
            float4 d2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 mine = b3 ? c4[j + 2] : c4[j], theirs = b3 ? c4[j] : c4[j + 2];
                d2[j].x = mine.x + ror8(theirs.x); d2[j].y = mine.y + ror8(theirs.y);
                d2[j].z = mine.z + ror8(theirs.z); d2[j].w = mine.w + ror8(theirs.w);
            }
            const float4 mine = b1 ? d2[1] : d2[0], theirs = b1 ? d2[0] : d2[1];
            float4 tot;
            tot.x = mine.x + qxor2(theirs.x); tot.y = mine.y + qxor2(theirs.y);
            tot.z = mine.z + qxor2(theirs.z); tot.w = mine.w + qxor2(theirs.w);
            tot.x += qxor1(tot.x); tot.y += qxor1(tot.y); tot.z += qxor1(tot.z); tot.w += qxor1(tot.w);
            int degk = 0;
            if (!p.backward && p.average) degk = bnd_of(rpv, krow + 1) - bnd_of(rpv, krow);
            if (v < n && (lane & 1) == 0) {
                if (p.self_loop) acc4(tot, self);
                if (!p.backward && p.average) {
                    const float d = (float)(degk + p.self_loop);   // 0/0 -> NaN as in the reference
                    tot.x /= d; tot.y /= d; tot.z /= d; tot.w /= d;
                }
                const int cc = col0 + 4 * q2;
                if (!p.self_loop) {
                    float4 sb = self;
                    if (prescale) {   // the tile holds dp/deg; the (1+eps) term needs dp itself
                        const float* src = p.x + (size_t)(row0 + v) * p.ldx + cc;
                        sb.x = (cc + 0 < p.F) ? src[0] : 0.f;
                        sb.y = (cc + 1 < p.F) ? src[1] : 0.f;
                        sb.z = (cc + 2 < p.F) ? src[2] : 0.f;
                        sb.w = (cc + 3 < p.F) ? src[3] : 0.f;
                    }
                    tot.x += selfB * sb.x; tot.y += selfB * sb.y; tot.z += selfB * sb.z; tot.w += selfB * sb.w;
                }
                float* dst = p.y + (size_t)(row0 + v) * p.ldy + cc;
                if (vec_out && cc + 3 < p.F) {
                    *reinterpret_cast<float4*>(dst) = tot;
                } else {
                    if (cc + 0 < p.F) dst[0] = tot.x;
                    if (cc + 1 < p.F) dst[1] = tot.y;
                    if (cc + 2 < p.F) dst[2] = tot.z;
                    if (cc + 3 < p.F) dst[3] = tot.w;
                }
            }
            rpv = nrpv;
        }
    } else {
    // y == null: only the d-eps dot product of phase A is wanted.
    // Nothing on the per-row path is a dependent global round trip: the row bounds come from LDS two rows ahead,
    // the row's first 64 column ids are requested one row ahead (unconditional loads: the arena keeps readable
    // slack behind the last block), and the ids of a chunk reach the lane groups either by DPP (8-lane groups: two
    // bank-masked row_newbcast moves, no LDS round trip) or by a BATCH of ds_bpermute issued before the reads.
    const unsigned jl2 = 2u * (unsigned)jlane;
    int v = p.y ? wave : n;
    int nb = 0, ne = 0, nnb = 0, nne = 0;
    unsigned nraw = 0;
    if (v < n) {
        nb = rp_s[v]; ne = rp_s[v + 1];
        nraw = load_id(cl, 2u * (unsigned)nb + jl2);
        if (v + nwaves < n) { nnb = rp_s[v + nwaves]; nne = rp_s[v + nwaves + 1]; }
    }
    for (; v < n; v += nwaves) {
        const int beg = nb, end = ne;
        unsigned raw = nraw;
        nb = nnb; ne = nne;
        if (v + nwaves < n) nraw = load_id(cl, 2u * (unsigned)nb + jl2);          // wave-uniform branch
        if (v + 2 * nwaves < n) { nnb = rp_s[v + 2 * nwaves]; nne = rp_s[v + 2 * nwaves + 1]; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e0 = beg; e0 < end; e0 += 64) {
            const int cnt = min(64, end - e0);
            if (e0 != beg) raw = load_id(cl, 2u * (unsigned)e0 + jl2);             // degree > 64: fetched in place
            const unsigned valb = (jlane < cnt) ? raw * (FS * 4) : zero_row_b;
            const int steps = (cnt + SLOTS - 1) / SLOTS;   // wave-uniform, <= LPR
            if constexpr (LPR == 8) {
                // padding slots read the zero row: blocks of 4 run unguarded, 4 reads in flight each
                {
                    const f32x4 t0 = lds_read16(bcast8<0>(valb) + subb), t1 = lds_read16(bcast8<1>(valb) + subb),
                                t2 = lds_read16(bcast8<2>(valb) + subb), t3 = lds_read16(bcast8<3>(valb) + subb);
                    acc4(acc, (t0 + t1) + (t2 + t3));
                }
                if (steps > 4) {
                    const f32x4 t0 = lds_read16(bcast8<4>(valb) + subb), t1 = lds_read16(bcast8<5>(valb) + subb),
                                t2 = lds_read16(bcast8<6>(valb) + subb), t3 = lds_read16(bcast8<7>(valb) + subb);
                    acc4(acc, (t0 + t1) + (t2 + t3));
                }
            } else {
                const int gbase = lane & ~(LPR - 1);
                int s = 0;
                for (; s + 4 <= steps; s += 4) {
                    const unsigned a0 = (unsigned)__shfl((int)valb, gbase + s, 64), a1 = (unsigned)__shfl((int)valb, gbase + s + 1, 64),
                                   a2 = (unsigned)__shfl((int)valb, gbase + s + 2, 64), a3 = (unsigned)__shfl((int)valb, gbase + s + 3, 64);
                    const f32x4 t0 = lds_read16(a0 + subb), t1 = lds_read16(a1 + subb), t2 = lds_read16(a2 + subb),
                                t3 = lds_read16(a3 + subb);
                    acc4(acc, (t0 + t1) + (t2 + t3));
                }
                for (; s < steps; ++s) acc4(acc, lds_read16((unsigned)__shfl((int)valb, gbase + s, 64) + subb));
            }
        }
        // combine the SLOTS partial sums of this destination row
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off, 64);
            acc.y += __shfl_xor(acc.y, off, 64);
            acc.z += __shfl_xor(acc.z, off, 64);
            acc.w += __shfl_xor(acc.w, off, 64);
        }
        if (lane < LPR) {
            const float4 self = tile[v * LPR + sub];
            if (p.self_loop) acc4(acc, self);
            if (!p.backward && p.average) {
                const float d = (float)(end - beg + p.self_loop);   // 0/0 -> NaN as in the reference
                acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d;
            }
            const int cc = col0 + 4 * sub;
            if (!p.self_loop) {
                float4 sb = self;
                if (prescale) {   // the tile holds dp/deg; the (1+eps) term needs dp itself
                    const float* src = p.x + (size_t)(row0 + v) * p.ldx + cc;
                    sb.x = (cc + 0 < p.F) ? src[0] : 0.f;
                    sb.y = (cc + 1 < p.F) ? src[1] : 0.f;
                    sb.z = (cc + 2 < p.F) ? src[2] : 0.f;
                    sb.w = (cc + 3 < p.F) ? src[3] : 0.f;
                }
                acc.x += selfB * sb.x; acc.y += selfB * sb.y; acc.z += selfB * sb.z; acc.w += selfB * sb.w;
            }
            float* dst = p.y + (size_t)(row0 + v) * p.ldy + cc;
            if (vec_out && cc + 3 < p.F) {
                *reinterpret_cast<float4*>(dst) = acc;
            } else {
                if (cc + 0 < p.F) dst[0] = acc.x;
                if (cc + 1 < p.F) dst[1] = acc.y;
                if (cc + 2 < p.F) dst[2] = acc.z;
                if (cc + 3 < p.F) dst[3] = acc.w;
            }
        }
    }

    }   // LPR != 8

    if (p.deps_partial) {   // block reduction of the d-eps dot product (fp64, fixed order)
        __syncthreads();    // everyone is done reading the tile; reuse its first bytes
        double* red = reinterpret_cast<double*>(smem);
        const double w = wave_sum_d(dot);
        if (lane == 0) red[wave] = w;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < nwaves; ++i) s += red[i];
            p.deps_partial[blockIdx.x] = s;
        }
    }
}


// ---------------------------------------------------------------------------------
// Specialised kernel for FS = 64 floats (LPR = 16, 256-B LDS rows): the headline shape
// (hidden_dim 64).  Same math as gnm_agg_kernel<16>; what changes is the schedule:
//   * phase A keeps 4 x 16 B (+ 4 x 16 B of hfwd) per thread in flight instead of one;
//   * the graph's rowptr is staged in LDS so the per-row bounds are LDS reads
//     (in order with the gather's ds_reads) instead of dependent scalar loads;
//   * waves take GROUPS OF 4 consecutive destination rows from an LDS ticket counter
//     (degree imbalance never idles a wave), and the column ids of the NEXT row are
//     fetched (two 64-id chunks) while the current row is gathered, so the global
//     index latency is off the critical path;
//   * gather steps run in unguarded blocks of 4 (padding slots point at a zero row),
//     so 4 ds_read_b128 are in flight before the first add waits;
//   * the four quarter-wave partial sums of 4 rows are combined by a 2-step
//     transposing butterfly (12 shuffles per 4 rows instead of 32), after which
//     quarter q owns row q: the epilogue and the 1-KiB output store use all 64 lanes.
// ---------------------------------------------------------------------------------
// One gather step of the 64-wide kernel reads 4 neighbour rows (one per 16-lane quarter).  Column ids arrive as
// aligned PAIRS (one dword per lane = two consecutive ids of a 128-id window): the low halves of the 64 lanes and
// the high halves each feed up to 16 steps, `ia` / `ib` being the LDS byte addresses they turned into.  A "double
// step" S reads step S of both.  Blocks of 4 double steps keep 8 ds_read_b128 in flight (32 registers).
#define GNM_RD(S) lds_read16(row_bcast16<S>(valb) + subb)
#define GNM_LO(S) lds_read16(row_bcast16<S>(ia) + subb)
#define GNM_HI(S) lds_read16(row_bcast16<S>(ib) + subb)
#define GNM_D1(S)                                                          \
    {                                                                      \
        const f32x4 a0_ = GNM_LO(S), b0_ = GNM_HI(S);                      \
        acc4(acc, a0_ + b0_);                                              \
    }
#define GNM_D2(S)                                                                                   \
    {                                                                                               \
        const f32x4 a0_ = GNM_LO(S), b0_ = GNM_HI(S), a1_ = GNM_LO(S + 1), b1_ = GNM_HI(S + 1);     \
        acc4(acc, (a0_ + b0_) + (a1_ + b1_));                                                       \
    }
#define GNM_D4W(S)                                                                                  \
    {                                                                                               \
        const f32x4 a0_ = GNM_LO(S), b0_ = GNM_HI(S), a1_ = GNM_LO(S + 1), b1_ = GNM_HI(S + 1),     \
                    a2_ = GNM_LO(S + 2), b2_ = GNM_HI(S + 2), a3_ = GNM_LO(S + 3), b3_ = GNM_HI(S + 3); \
        acc4(acc, ((a0_ + b0_) + (a1_ + b1_)) + ((a2_ + b2_) + (a3_ + b3_)));                       \
    }
// (the STATS kernel carries ~25 more live registers through the gather: 4 reads in flight per block there)
#define GNM_D4(S)                      \
    if constexpr (STATS) {             \
        GNM_D2(S)                      \
        GNM_D2((S) + 2)                \
    } else {                           \
        GNM_D4W(S)                     \
    }
// t in 0..3 double steps from step B on, exactly
#define GNM_DTAIL(B, t)                       \
    if ((t) >= 2) {                           \
        GNM_D2(B)                             \
        if ((t) >= 3) GNM_D1((B) + 2)         \
    } else if ((t) >= 1) {                    \
        GNM_D1(B)                             \
    }
// 8 double steps = 16 reads in flight (64 registers: the forward kernel only)
#define GNM_D8W(S)                                                                                  \
    {                                                                                               \
        const f32x4 a0_ = GNM_LO(S), b0_ = GNM_HI(S), a1_ = GNM_LO(S + 1), b1_ = GNM_HI(S + 1),     \
                    a2_ = GNM_LO(S + 2), b2_ = GNM_HI(S + 2), a3_ = GNM_LO(S + 3), b3_ = GNM_HI(S + 3), \
                    a4_ = GNM_LO(S + 4), b4_ = GNM_HI(S + 4), a5_ = GNM_LO(S + 5), b5_ = GNM_HI(S + 5), \
                    a6_ = GNM_LO(S + 6), b6_ = GNM_HI(S + 6), a7_ = GNM_LO(S + 7), b7_ = GNM_HI(S + 7); \
        acc4(acc, ((a0_ + b0_) + (a1_ + b1_)) + ((a2_ + b2_) + (a3_ + b3_)));                       \
        acc4(acc, ((a4_ + b4_) + (a5_ + b5_)) + ((a6_ + b6_) + (a7_ + b7_)));                       \
    }
#ifndef GNM_LADDER_MODE
#define GNM_LADDER_MODE 0
#endif
// nd in 0..16 double steps.  Mode 0: exactly nd, in blocks of 4 + a tail of 2 / 1.  Tuning variants: 1 = first 8 as
// one 16-deep block; 2 = nd rounded up to whole blocks of 4 (padding reads the zero row: no tail pieces);
// 3 = both.
#define GNM_DLADDER_EXACT(nd)                             \
    if ((nd) >= 4) {                                      \
        GNM_D4(0)                                         \
        if ((nd) >= 8) {                                  \
            GNM_D4(4)                                     \
            if ((nd) >= 12) {                             \
                GNM_D4(8)                                 \
                if ((nd) >= 16) {                         \
                    GNM_D4(12)                            \
                } else {                                  \
                    GNM_DTAIL(12, (nd) - 12)              \
                }                                         \
            } else {                                      \
                GNM_DTAIL(8, (nd) - 8)                    \
            }                                             \
        } else {                                          \
            GNM_DTAIL(4, (nd) - 4)                        \
        }                                                 \
    } else {                                              \
        GNM_DTAIL(0, nd)                                  \
    }
#define GNM_DLADDER_D8(nd)                                \
    if (!STATS && (nd) >= 8) {                            \
        GNM_D8W(0)                                        \
        if ((nd) >= 12) {                                 \
            GNM_D4(8)                                     \
            if ((nd) >= 16) {                             \
                GNM_D4(12)                                \
            } else {                                      \
                GNM_DTAIL(12, (nd) - 12)                  \
            }                                             \
        } else {                                          \
            GNM_DTAIL(8, (nd) - 8)                        \
        }                                                 \
    } else {                                              \
        GNM_DLADDER_EXACT(nd)                             \
    }
#define GNM_DLADDER_PAD(nd)                               \
    if ((nd) > 0) {                                       \
        GNM_D4(0)                                         \
        if ((nd) > 4) {                                   \
            GNM_D4(4)                                     \
            if ((nd) > 8) {                               \
                GNM_D4(8)                                 \
                if ((nd) > 12) GNM_D4(12)                 \
            }                                             \
        }                                                 \
    }
#define GNM_DLADDER_D8PAD(nd)                             \
    if (!STATS && (nd) > 4) {                             \
        GNM_D8W(0)                                        \
        if ((nd) > 8) {                                   \
            if ((nd) > 12) {                              \
                GNM_D8W(8)                                \
            } else {                                      \
                GNM_D4(8)                                 \
            }                                             \
        }                                                 \
    } else {                                              \
        GNM_DLADDER_PAD(nd)                               \
    }
#if GNM_LADDER_MODE == 1
#define GNM_DLADDER(nd) GNM_DLADDER_D8(nd)
#elif GNM_LADDER_MODE == 2
#define GNM_DLADDER(nd) GNM_DLADDER_PAD(nd)
#elif GNM_LADDER_MODE == 3
#define GNM_DLADDER(nd) GNM_DLADDER_D8PAD(nd)
#else
#define GNM_DLADDER(nd) GNM_DLADDER_EXACT(nd)
#endif
// single-register forms (ids fetched in place beyond the 128-id window: degree > 128)
#define GNM_S1(S) acc4(acc, GNM_RD(S));
#define GNM_S2(S)                                                \
    {                                                            \
        const f32x4 p0_ = GNM_RD(S), p1_ = GNM_RD((S) + 1);      \
        acc4(acc, p0_ + p1_);                                    \
    }
#define GNM_S4(S)                                                                                          \
    {                                                                                                      \
        const f32x4 t0_ = GNM_RD(S), t1_ = GNM_RD((S) + 1), t2_ = GNM_RD((S) + 2), t3_ = GNM_RD((S) + 3);  \
        acc4(acc, (t0_ + t1_) + (t2_ + t3_));                                                              \
    }
#define GNM_S8(S)                                                                                          \
    {                                                                                                      \
        const f32x4 a0_ = GNM_RD(S), a1_ = GNM_RD((S) + 1), a2_ = GNM_RD((S) + 2), a3_ = GNM_RD((S) + 3),  \
                    a4_ = GNM_RD((S) + 4), a5_ = GNM_RD((S) + 5), a6_ = GNM_RD((S) + 6), a7_ = GNM_RD((S) + 7); \
        acc4(acc, ((a0_ + a1_) + (a2_ + a3_)) + ((a4_ + a5_) + (a6_ + a7_)));                              \
    }
// t in 0..8 steps from step B on, exactly
#define GNM_STAIL(B, t)                              \
    if ((t) >= 8) {                                  \
        GNM_S8(B)                                    \
    } else if ((t) >= 4) {                           \
        GNM_S4(B)                                    \
        if ((t) >= 6) {                              \
            GNM_S2((B) + 4)                          \
            if ((t) >= 7) GNM_S1((B) + 6)            \
        } else if ((t) >= 5) {                       \
            GNM_S1((B) + 4)                          \
        }                                            \
    } else if ((t) >= 2) {                           \
        GNM_S2(B)                                    \
        if ((t) >= 3) GNM_S1((B) + 2)                \
    } else if ((t) >= 1) {                           \
        GNM_S1(B)                                    \
    }
#define GNM_SLADDER(ns)                  \
    if ((ns) > 8) {                      \
        GNM_S8(0)                        \
        GNM_STAIL(8, (ns) - 8)           \
    } else {                             \
        GNM_STAIL(0, ns)                 \
    }
#define GNM_BLOCK16()                                                                                   \
    {                                                                                                   \
        GNM_S8(0)                                                                                       \
        GNM_S8(8)                                                                                       \
    }


__device__ __forceinline__ float4 sel4(bool c, const float4 a, const float4 b) {
    return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}
__device__ __forceinline__ float4 shfl_xor4(const float4 v, int m) {
    return make_float4(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64), __shfl_xor(v.z, m, 64),
                       __shfl_xor(v.w, m, 64));
}

template <bool STATS>
__global__ void __launch_bounds__(1024) gnm_agg16_kernel(const AggArgs p) {
    constexpr int LPR = 16;
    constexpr int FS = 64;
#ifdef GNM_AGG16_TUNING       // tools/bench_agg.py ablations (GNM_AGG16_DEBUG); compiled out of the product kernel
    const int dbg = p.debug;
    // in-kernel timeline (tools/agg_timeline.py): lane 0 of every wave drops s_memtime stamps at phase / group
    // boundaries, where no LDS read is in flight anyway (s_memtime returns through lgkmcnt)
#define GNM_STAMP(k)                                                                                          \
    if (p.stamps && (threadIdx.x & 63) == 0)                                                                  \
        p.stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 64 + (k)] = __builtin_amdgcn_s_memtime();
#else
    constexpr int dbg = 0;
#define GNM_STAMP(k)
#endif
    GNM_STAMP(0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* tile = reinterpret_cast<float4*>(smem);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int b = blockIdx.x / p.nslices;
    const int sl = blockIdx.x - b * p.nslices;
    const int row0 = p.node_off[b];
    const int n = p.node_off[b + 1] - row0;
    const int col0 = sl * FS;
    const int32_t* rp = p.rowptr + p.b_rp_off[b];
    const uint16_t* cl = p.col + p.b_col_off[b];
    const int32_t* drp = p.deg_rowptr + p.b_deg_off[b];
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const bool vec_in = ((p.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
    const bool vec_h = p.hfwd && ((p.ldh & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.hfwd) & 15) == 0);
    const bool prescale = p.backward && p.average;
    const int nwaves = nthreads >> 6;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane & 15;
    const int q = lane >> 4;                                   // quarter of the wave
    const int jlane = sub * 4 + q;                             // edge of a 64-chunk whose id this lane fetches
    const unsigned jl2 = 2u * (unsigned)jlane;

    // Row schedule: the rows are cut into GROUPS -- 4 consecutive rows each, except that the last `tail_rows` rows
    // (about one per wave) are groups of ONE row -- and the groups are handed to the waves by TICKET (an LDS
    // counter), three groups ahead of the gather:  ticket -> row bounds (global load) -> column ids (global load) ->
    // gather, one pipeline stage per group.  A static deal left every wave idle for 16 % of the workgroup's time
    // on average, waiting for the workgroup's slowest wave (in-kernel timeline, profiles/r02_agg_timeline.md): the
    // waves see the same row count but not the same LDS service.  The first three tickets of a wave are fixed
    // (wave, wave + W, wave + 2W), so the pipeline is primed without a round trip.
    // The kernel that also reduces column statistics over the rows (STATS) keeps a STATIC deal instead -- full rounds
    // of 4-row groups round-robin over the waves, the remainder as one short group per wave -- because those sums
    // are accumulated per wave: with tickets their summation order, hence their last bits, would change from launch
    // to launch (every reduction in this library has a fixed order; tests/test_gpu_model_parity.py
    // test_run_to_run_determinism).  Row results themselves do not depend on which wave gathers them.
    constexpr bool DYN = !STATS;
    const int tail_rows = n >= nwaves + 3 ? nwaves + ((n - nwaves) & 3) : n;
    const int n4 = (n - tail_rows) >> 2;                       // DYN: 4-row groups = tickets 0 .. n4-1; then single rows
    const int nfull = n / (4 * nwaves);                        // static: full rounds, then one short group per wave
    const int trem = n - nfull * 4 * nwaves;
    const int tbase = trem / nwaves, textra = trem - tbase * nwaves;
    const int tfirst = nfull * 4 * nwaves + wave * tbase + min(wave, textra);
    const int trows = tbase + (wave < textra ? 1 : 0);
    // tickets >= gtot: nothing left.  DYN: tickets number the groups of the graph; static: the groups of this wave
    const int gtot = !p.y ? 0 : (DYN ? n4 + tail_rows : nfull + (trem > 0 ? 1 : 0));
    int* const ticket = reinterpret_cast<int*>(smem + (size_t)(n + 1) * (FS * 4));     // LDS word behind the tile
    // bounds of group t as a lane vector: lane i (0..4) = rowptr[first row + min(i, rows of the group)], read
    // straight from global memory (a 1.6 KB array, L2-resident); an exhausted ticket yields an empty group at row 0
    auto group_first = [&](int t) -> int {
        if constexpr (DYN) return t < n4 ? 4 * t : (t < gtot ? 4 * n4 + (t - n4) : 0);
        else return t < nfull ? 4 * (wave + t * nwaves) : (t < gtot ? tfirst : 0);
    };
    auto group_rows = [&](int t) -> int {
        if constexpr (DYN) return t < n4 ? 4 : (t < gtot ? 1 : 0);
        else return t < nfull ? 4 : (t < gtot ? trows : 0);
    };
    auto load_bounds = [&](int t) -> int { return rp[min(group_first(t) + min(lane & 7, group_rows(t)), n)]; };
    // Column ids are requested FOUR rows (one group) ahead into a ring of 4 x 2 registers, as RAW values that are
    // only turned into LDS addresses when their row becomes current.  vmcnt completes in issue order, stores
    // included: with a one-row lookahead every wait for ids also waited for the output store issued just before
    // them; now a wait only ever targets loads that are a whole group old.
    // One request = one aligned DWORD per lane = two consecutive ids: the 64 lanes cover a 128-id window that starts
    // at the row's first id rounded down to an even position of p.col (so any b_col_off works); a second request
    // covers the next 128 ids (15 % of the rows of the 400-node benchmark graphs have more than 128 neighbours:
    // fetched in place, each cost a dependent round trip and a drained queue).  The loads are unconditional (p.col
    // stays readable 256 ids past the last block): a load inside a divergent branch would be drained at the join.
    const long long colbase = p.b_col_off[b];
    const uint16_t* clp = p.col + (colbase & ~1LL);            // 4-byte aligned (launcher checks p.col)
    const int cpar = (int)(colbase & 1);
    const unsigned jl4 = 4u * (unsigned)jlane;
    auto load_pair = [&](int beg, unsigned window) -> unsigned {    // window w of the row that starts at local id `beg`
        const unsigned a0 = (unsigned)(cpar + beg) & ~1u;
        return *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(clp) + ((a0 << 1) + 256u * window + jl4));
    };
    // The first two bounds vectors are requested here, unconditionally; the ring is filled by issue_ring() at exactly
    // ONE place on every path through phase A -- after the first batch of tile loads is in flight and before
    // anything is stored.  (A request that is pending on a path the compiler cannot rule out becomes a vmcnt(0)
    // drain at its next use, inside the group loop.)
    int tA = DYN ? wave : 0, tB = DYN ? wave + nwaves : 1, tC = DYN ? wave + 2 * nwaves : 2;
    int bvA = load_bounds(tA), bvB = load_bounds(tB);
    unsigned idw[4], idx[4];
    auto issue_ring = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int beg = __builtin_amdgcn_readlane(bvA, r);
            idw[r] = load_pair(beg, 0);
            idx[r] = load_pair(beg, 1);
        }
    };
    const float selfB = p.self_loop ? 0.f : (p.eps ? 1.f + *p.eps : 1.f);

    // ---- phase A ---------------------------------------------------------------
    double dot = 0.0;
    const int total = n * LPR;
    constexpr int UNR = 4;
    // d eps: either dot(x, hfwd) on the way in (hfwd given), or -- STATS launches -- from the rows the epilogue
    // already holds: x's own row and h recomputed from the layer below's Z (no pass over hfwd at all)
    const bool dot_a = p.deps_partial && p.hfwd;
    const bool fast = vec_in && (col0 + FS <= p.F) && (!dot_a || vec_h);
    int base = tid;
    // forward prologue (non-STATS launches of gnm_agg_fwd_bnrelu only; nthreads is a multiple of 16, so a
    // thread keeps the column chunk tid & 15 for all its rows)
    const bool pro = !STATS && p.p_scale != nullptr;
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pro) {
        psc = *reinterpret_cast<const float4*>(p.p_scale + 4 * (tid & 15));
        psh = *reinterpret_cast<const float4*>(p.p_shift + 4 * (tid & 15));
        auto emit = [&](int i, float4 w) {
            w.x = gnm_relu(w.x * psc.x + psh.x); w.y = gnm_relu(w.y * psc.y + psh.y);
            w.z = gnm_relu(w.z * psc.z + psh.z); w.w = gnm_relu(w.w * psc.w + psh.w);
            if (p.p_hout) *reinterpret_cast<float4*>(p.p_hout + (size_t)(row0 + (i >> 4)) * p.p_ldh + 4 * (i & 15)) = w;
            csum.x += w.x; csum.y += w.y; csum.z += w.z; csum.w += w.w;
            tile[i] = w;
        };
        if (base + (UNR - 1) * nthreads < total) {      // first batch, peeled: the id ring is requested behind its loads
            float4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = base + u * nthreads;
                v[u] = *reinterpret_cast<const float4*>(p.x + (size_t)(row0 + (i >> 4)) * p.ldx + 4 * (i & 15));
            }
            issue_ring();
#pragma unroll
            for (int u = 0; u < UNR; ++u) emit(base + u * nthreads, v[u]);
            base += nthreads * UNR;
        } else {
            issue_ring();
        }
        for (; base + (UNR - 1) * nthreads < total; base += nthreads * UNR) {     // UNR x 16 B per thread in flight
            float4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = base + u * nthreads;
                v[u] = *reinterpret_cast<const float4*>(p.x + (size_t)(row0 + (i >> 4)) * p.ldx + 4 * (i & 15));
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) emit(base + u * nthreads, v[u]);
        }
        {
            float4 v[UNR];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = base + u * nthreads;
                if (i < total) {
                    v[u] = *reinterpret_cast<const float4*>(p.x + (size_t)(row0 + (i >> 4)) * p.ldx + 4 * (i & 15));
                    cnt = u + 1;
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (u < cnt) emit(base + u * nthreads, v[u]);
            base = total;
        }
        // readout partials: [nthreads] float4 behind the row offsets (the 64 threads of a column chunk)
        float4* rsum = reinterpret_cast<float4*>(smem + (((size_t)(n + 1) * (FS * 4) + (size_t)(n + 2) * 4 + 15) & ~(size_t)15));
        rsum[tid] = csum;
    } else {
    // branch-free main part: UNR x 16 B (+ UNR x 16 B of hfwd) per thread in flight; the id ring is requested
    // behind the loads of the first batch
    auto fast_batch = [&](int base, bool with_ring) {
        float4 v[UNR], hh[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = base + u * nthreads;
            const size_t off = (size_t)(row0 + (i >> 4)) * p.ldx + col0 + 4 * (i & 15);
            v[u] = *reinterpret_cast<const float4*>(p.x + off);
        }
        if (dot_a) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = base + u * nthreads;
                const size_t off = (size_t)(row0 + (i >> 4)) * p.ldh + col0 + 4 * (i & 15);
                hh[u] = *reinterpret_cast<const float4*>(p.hfwd + off);
            }
        }
        if (with_ring) issue_ring();
        if (dot_a) {
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                dot += (double)v[u].x * hh[u].x + (double)v[u].y * hh[u].y + (double)v[u].z * hh[u].z +
                       (double)v[u].w * hh[u].w;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = base + u * nthreads;
            float4 w = v[u];
            if (prescale) {
                const int r = i >> 4;
                const float d = (float)(drp[r + 1] - drp[r] + p.self_loop);
                w.x /= d; w.y /= d; w.z /= d; w.w /= d;
            }
            tile[i] = w;
        }
    };
    if (fast && base + (UNR - 1) * nthreads < total) {
        fast_batch(base, true);
        base += nthreads * UNR;
        for (; base + (UNR - 1) * nthreads < total; base += nthreads * UNR) fast_batch(base, false);
    } else {
        issue_ring();
    }
    for (; base < total; base += nthreads) {      // tail, and the generic (unaligned / partial-width) path
        const int i = base;
        const int r = i >> 4, c = i & 15;
        const int cc = col0 + 4 * c;
        const float* src = p.x + (size_t)(row0 + r) * p.ldx + cc;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec_in && cc + 3 < p.F) {
            w = *reinterpret_cast<const float4*>(src);
        } else {
            if (cc + 0 < p.F) w.x = src[0];
            if (cc + 1 < p.F) w.y = src[1];
            if (cc + 2 < p.F) w.z = src[2];
            if (cc + 3 < p.F) w.w = src[3];
        }
        if (dot_a) {
            const float* hs = p.hfwd + (size_t)(row0 + r) * p.ldh + cc;
            float4 h4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vec_h && cc + 3 < p.F) {
                h4 = *reinterpret_cast<const float4*>(hs);
            } else {
                if (cc + 0 < p.F) h4.x = hs[0];
                if (cc + 1 < p.F) h4.y = hs[1];
                if (cc + 2 < p.F) h4.z = hs[2];
                if (cc + 3 < p.F) h4.w = hs[3];
            }
            dot += (double)w.x * h4.x + (double)w.y * h4.y + (double)w.z * h4.z + (double)w.w * h4.w;
        }
        if (prescale) {
            const float d = (float)(drp[r + 1] - drp[r] + p.self_loop);
            w.x /= d; w.y /= d; w.z /= d; w.w /= d;
        }
        tile[i] = w;
    }
    }
    // fused BatchNorm-backward statistics of the layer below (STATS): per-lane column chunk `sub` (requested in
    // front of the drain below, like everything else the group loop finds already in registers)
    float4 ss1 = make_float4(0.f, 0.f, 0.f, 0.f), ss2 = ss1, s_pb = ss1, s_ub = ss1;
    float4 lsc = ss1, lsh = ss1, lmu = ss1;
    if constexpr (STATS) {
        lsc = *reinterpret_cast<const float4*>(p.s_scale + 4 * sub);
        lsh = *reinterpret_cast<const float4*>(p.s_shift + 4 * sub);
        lmu = *reinterpret_cast<const float4*>(p.s_mean + 4 * sub);      // (rstd multiplies the column sums at the end)
        if (p.s_dpool) {
            s_pb = *reinterpret_cast<const float4*>(p.s_dpool + (size_t)b * p.ld_dpool + 4 * sub);
            if (p.s_avg) {
                const float w = 1.0f / (float)n;
                s_pb.x *= w; s_pb.y *= w; s_pb.z *= w; s_pb.w *= w;
            }
        }
        if (p.s_dsc1) s_ub = *reinterpret_cast<const float4*>(p.s_U + (size_t)b * p.ld_U + 4 * sub);
    }

    if (tid < LPR) tile[n * LPR + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    // Drain this wave's vector-memory queue HERE, once per graph, in front of the barrier (where all but the last
    // wave wait anyway): the group loop below then starts from a known-empty queue and its waits come out as the
    // exact counted vmcnt(3) per row.  Without it the compiler merges the several phase-A paths into "something may
    // be pending" and drains the queue -- output stores included -- at the top of every group.
    if (DYN && tid == 0) *ticket = 3 * nwaves;   // tickets 0 .. 3W-1 are the fixed first three of every wave
    GNM_STAMP(1)
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0); expcnt / lgkmcnt left alone
    GNM_STAMP(2)
    __syncthreads();
    GNM_STAMP(3)
    if (pro && p.p_gf && tid < LPR) {
        // graph readout of the activation just formed (graphcnn.py:229), partials combined in thread order
        const float4* rsum = reinterpret_cast<const float4*>(smem + (((size_t)(n + 1) * (FS * 4) + (size_t)(n + 2) * 4 + 15) & ~(size_t)15));
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = tid; k < nthreads; k += LPR) acc4(t, rsum[k]);
        if (p.p_gf_avg) {
            const float inv = 1.f / (float)n;
            t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
        }
        *reinterpret_cast<float4*>(p.p_gf + (size_t)b * p.p_ldgf + 4 * tid) = t;
    }

    // ---- phase B ---------------------------------------------------------------
    const unsigned subb = lds_base + (unsigned)sub * 16u;
    const unsigned zero_row_b = (unsigned)n * (FS * 4);
    const bool vec_out = ((p.ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0);
    const bool vec_full = vec_out && (col0 + FS <= p.F);       // this slice is full width and 16-byte addressable
    const bool need_deg = !STATS && !p.backward && p.average;  // (STATS launches are backward launches)
    const int qm1 = -(q & 1), qm2 = -(q >> 1);               // all-ones masks selecting this lane's quarter
    // No LDS round trip sits on the per-row critical path: under load the LDS queue is ~1000 cycles deep, so row
    // bounds come from the lane vector (v_readlane), column ids arrive a group ahead, and the 4-row combine below
    // is pure VALU (v_permlane32_swap / v_permlane16_swap).
    if (dbg & 8) {   // tuning: the micro-benchmark's steady-state loop, same step count (2 x 16 per row)
        unsigned valb = (unsigned)((lane * 37 + 11) % max(n, 1)) * (FS * 4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < (n + 4 * nwaves - 1) / (4 * nwaves); ++g)
            for (int r = 0; r < 8; ++r) {
                GNM_BLOCK16()
                valb ^= 256;
            }
        if (acc.x == 12345.678f) p.y[0] = acc.y + acc.z + acc.w;
        return;
    }
    int kk = 0;
    while (tA < gtot) {                                                        // wave-uniform
        GNM_STAMP(4 + 5 * min(kk, 10))
        // ticket for the group three ahead (one LDS atomic by lane 0, read at the bottom of this iteration, by
        // which time every LDS operation issued before it has long returned), and its predecessor's bounds
        int tD = tC + 1;
        if constexpr (DYN) {
            if (lane == 0) tD = atomicAdd(ticket, 1);
        }
        const int bvC = load_bounds(tC);
        const int gfirst = group_first(tA);
        const int grows = group_rows(tA);                                      // rows of this group (quarter q owns row q)
        // this quarter's own row (for the self term): read it NOW, ahead of the gather, so it
        // is not a dependent LDS round trip behind ~256 queued reads in the epilogue
        const int v = gfirst + q;
        const float4 self = tile[min(v, n) * LPR + sub];
        // "average" backward: the tile holds x / deg, the (1 + eps) self term needs x itself -- requested here,
        // ahead of the gather (a load issued in the epilogue is waited for at once, which drains the whole queue)
        float4 xself = make_float4(0.f, 0.f, 0.f, 0.f);
        if (prescale && !p.self_loop) {                                  // wave-uniform
            const int cc = col0 + 4 * sub;
            const float* src = p.x + (size_t)(row0 + min(v, n - 1)) * p.ldx + cc;
            if (vec_in && col0 + FS <= p.F) {
                xself = *reinterpret_cast<const float4*>(src);
            } else {
                xself.x = (cc + 0 < p.F) ? src[0] : 0.f;
                xself.y = (cc + 1 < p.F) ? src[1] : 0.f;
                xself.z = (cc + 2 < p.F) ? src[2] : 0.f;
                xself.w = (cc + 3 < p.F) ? src[3] : 0.f;
            }
        }
        float4 zrow = make_float4(0.f, 0.f, 0.f, 0.f);
        float dsc_v = 0.f;
        if constexpr (STATS) {     // requested ahead of the gather, consumed in the epilogue
            const int vc = row0 + min(v, n - 1);
            zrow = *reinterpret_cast<const float4*>(p.sZ + (size_t)vc * p.ldsz + 4 * sub);
            if (p.s_dsc1) dsc_v = p.s_dsc1[vc];
        }
        // row bounds of this group
        const int b0 = __builtin_amdgcn_readlane(bvA, 0), b1 = __builtin_amdgcn_readlane(bvA, 1),
                  b2 = __builtin_amdgcn_readlane(bvA, 2), b3 = __builtin_amdgcn_readlane(bvA, 3),
                  b4 = __builtin_amdgcn_readlane(bvA, 4);
        // degree of row v (forward "average" only): a branch-free per-quarter select (written as a ?: chain it
        // compiled to ~40 exec-mask/branch instructions per group, on every launch)
        int degv = 0;
        if (need_deg) {                                    // wave-uniform
            const int d0 = b1 - b0, d1 = b2 - b1, d2 = b3 - b2, d3 = b4 - b3;
            const int lo = (d0 & ~qm1) | (d1 & qm1), hi = (d2 & ~qm1) | (d3 & qm1);
            degv = (lo & ~qm2) | (hi & qm2);
        }
        float4 racc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int beg = r == 0 ? b0 : (r == 1 ? b1 : (r == 2 ? b2 : b3));
            const int end = r == 0 ? b1 : (r == 1 ? b2 : (r == 2 ? b3 : b4));
            const int cnt = end - beg;
            // This row's first window becomes LDS addresses FIRST (the raw value is then dead), and only then is
            // its ring slot refilled: written the other way round the compiler parks all four old values in copies
            // at the top of the group, which needs every outstanding request -- and the store -- complete.
            // Window position of an id = 2 * jlane (+ 1 for the high half); the row's ids sit at positions
            // o .. o + cnt - 1 for o = parity of the row start; the rest read the zero row.
            const int o = (cpar + beg) & 1;
            const unsigned wl = 2u * (unsigned)jlane - (unsigned)o;
            unsigned ia = (wl < (unsigned)cnt) ? ((idw[r] & 0xffffu) << 8) : zero_row_b;
            unsigned ib = (wl + 1u < (unsigned)cnt) ? ((idw[r] >> 16) << 8) : zero_row_b;
            // (compiler barrier: keeps the two address computations above the refill -- left alone, they are sunk
            // into the gather below it)
            asm volatile("" : "+v"(ia), "+v"(ib) : : "memory");
            // the windows of row r of the NEXT group (an exhausted ticket: requests at a valid address whose
            // results are never used)
            const int nbeg = __builtin_amdgcn_readlane(bvB, r);
            if (!(dbg & 1)) idw[r] = load_pair(nbeg, 0);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cnt > 0) {                                 // wave-uniform
                const int used = cnt + o;                            // window positions [o, used) hold this row's ids
                const int nd = min((used + 7) >> 3, 16);            // double steps (4 low + 4 high ids each)
                GNM_DLADDER(nd)
                if (used > 128) {                                    // second window (wave-uniform)
                    const unsigned wl2 = wl + 128u;
                    unsigned ia2 = (wl2 < (unsigned)cnt) ? ((idx[r] & 0xffffu) << 8) : zero_row_b;
                    unsigned ib2 = (wl2 + 1u < (unsigned)cnt) ? ((idx[r] >> 16) << 8) : zero_row_b;
                    asm volatile("" : "+v"(ia2), "+v"(ib2) : : "memory");
                    {
                        const unsigned ia = ia2, ib = ib2;
                        const int nd2 = min((used - 128 + 7) >> 3, 16);
                        GNM_DLADDER(nd2)
                    }
                    // degree beyond both windows (> 256 - o ids): 64 ids at a time, fetched in place
                    for (int e0 = beg + 256 - o; e0 < end; e0 += 64) {
                        const int c64 = min(end - e0, 64);
                        const unsigned rc = load_id(cl, 2u * (unsigned)e0 + jl2);
                        const unsigned valb = (jlane < c64) ? rc * (FS * 4) : zero_row_b;
                        const int ns = (c64 + 3) >> 2;
                        GNM_SLADDER(ns)
                    }
                }
            }
            if (!(dbg & 1)) idx[r] = load_pair(nbeg, 1);
            racc[r] = acc;
        }
        GNM_STAMP(5 + 5 * min(kk, 10))
        // transposing combine: afterwards quarter q holds the full sum of row q of the group
        float4 tot;
        if (dbg & 4) {
            tot = racc[0]; acc4(tot, racc[1]); acc4(tot, racc[2]); acc4(tot, racc[3]);
        } else {
            const float4 t02 = swap_add32(racc[0], racc[2]);   // lanes 0-31: row 0, lanes 32-63: row 2
            const float4 t13 = swap_add32(racc[1], racc[3]);
            tot = swap_add16(t02, t13);
        }
        GNM_STAMP(6 + 5 * min(kk, 10))

        if (dbg & 2) {
            if (tot.x == 12345.678f) p.y[0] = tot.y + tot.z + tot.w + self.x;   // keep values live
        } else if (q < grows) {
            if (p.self_loop) acc4(tot, self);
            if (need_deg) {
                const float d = (float)(degv + p.self_loop);   // 0/0 -> NaN as in the reference
                tot.x /= d; tot.y /= d; tot.z /= d; tot.w /= d;
            }
            const int cc = col0 + 4 * sub;
            if (!p.self_loop) {
                const float4 sb = prescale ? xself : self;
                tot.x += selfB * sb.x; tot.y += selfB * sb.y; tot.z += selfB * sb.z; tot.w += selfB * sb.w;
                if constexpr (STATS) {
                    if (p.deps_partial && !p.hfwd) {
                        // d eps += dpooled[v] . h[v], h = relu(bn_lo(Z[v])) recomputed as the forward formed it
                        const float hx = gnm_relu(zrow.x * lsc.x + lsh.x), hy = gnm_relu(zrow.y * lsc.y + lsh.y);
                        const float hz = gnm_relu(zrow.z * lsc.z + lsh.z), hw = gnm_relu(zrow.w * lsc.w + lsh.w);
                        dot += (double)(sb.x * hx + sb.y * hy) + (double)(sb.z * hz + sb.w * hw);
                    }
                }
            }
            if constexpr (STATS) {
                // total gradient at this layer output = aggregation backward + readout + discriminator terms
                tot.x += s_pb.x + dsc_v * s_ub.x; tot.y += s_pb.y + dsc_v * s_ub.y;
                tot.z += s_pb.z + dsc_v * s_ub.z; tot.w += s_pb.w + dsc_v * s_ub.w;
                if (p.s_dsc1 && row0 + v < p.n_batch) {     // rows perm[g] < B of the shuffled branch (graphcnn.py:242)
                    const int gq = gnm_perm_entry(p.s_inv_perm[row0 + v], p.n_batch);
                    const float s2 = p.s_s2sum[gq];
                    const float4 uq = *reinterpret_cast<const float4*>(p.s_U + (size_t)gq * p.ld_U + 4 * sub);
                    tot.x += s2 * uq.x; tot.y += s2 * uq.y; tot.z += s2 * uq.z; tot.w += s2 * uq.w;
                }
                if (!(zrow.x * lsc.x + lsh.x > 0.f)) tot.x = 0.f;
                if (!(zrow.y * lsc.y + lsh.y > 0.f)) tot.y = 0.f;
                if (!(zrow.z * lsc.z + lsh.z > 0.f)) tot.z = 0.f;
                if (!(zrow.w * lsc.w + lsh.w > 0.f)) tot.w = 0.f;
                ss1.x += tot.x; ss1.y += tot.y; ss1.z += tot.z; ss1.w += tot.w;
                ss2.x += tot.x * (zrow.x - lmu.x); ss2.y += tot.y * (zrow.y - lmu.y);
                ss2.z += tot.z * (zrow.z - lmu.z); ss2.w += tot.w * (zrow.w - lmu.w);
            }
            GNM_STAMP(7 + 5 * min(kk, 10))
            float* dst = p.y + (size_t)(row0 + v) * p.ldy + cc;
            if (vec_full) {     // wave-uniform: ONE 16-byte store per lane (a per-lane condition here was
                                // if-converted into a 12-byte + a 4-byte masked store on every launch)
                // (a plain float4 store here is merged by the optimiser with the scalar stores of the other
                //  branch into a 12-byte + a 4-byte store; the streaming flavour cannot be: ONE
                //  global_store_dwordx4 nt.  The row is written once and next read by another kernel.)
                const f32x4 t4 = {tot.x, tot.y, tot.z, tot.w};
                __builtin_nontemporal_store(t4, reinterpret_cast<f32x4*>(dst));
            } else {
                if (cc + 0 < p.F) dst[0] = tot.x;
                if (cc + 1 < p.F) dst[1] = tot.y;
                if (cc + 2 < p.F) dst[2] = tot.z;
                if (cc + 3 < p.F) dst[3] = tot.w;
            }
        }
        GNM_STAMP(8 + 5 * min(kk, 10))
        // advance the pipeline
        tA = tB; tB = tC; tC = __builtin_amdgcn_readfirstlane(tD);
        bvA = bvB; bvB = bvC;
        ++kk;
    }
    GNM_STAMP(63)

    if constexpr (STATS) {     // column sums: the 4 quarters of a wave, then the waves, in a fixed order
        __syncthreads();
        double* sred = reinterpret_cast<double*>(smem) + 64;      // [nwaves][2][64] (after the d-eps slots)
        const float v1[4] = {ss1.x, ss1.y, ss1.z, ss1.w}, v2[4] = {ss2.x, ss2.y, ss2.z, ss2.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double d1 = (double)v1[c], d2 = (double)v2[c];
            d1 += __shfl_xor(d1, 16, 64); d2 += __shfl_xor(d2, 16, 64);
            d1 += __shfl_xor(d1, 32, 64); d2 += __shfl_xor(d2, 32, 64);
            if (lane < 16) {
                sred[(wave * 2 + 0) * 64 + 4 * sub + c] = d1;
                sred[(wave * 2 + 1) * 64 + 4 * sub + c] = d2;
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, col = tid & 63;
            double sum = 0.0;
            for (int w = 0; w < nwaves; ++w) sum += sred[(w * 2 + which) * 64 + col];
            if (which) sum *= (double)p.s_rstd[col];      // sum G (Z - mean) -> sum G xhat
            p.s_partial[((size_t)blockIdx.x * 2 + which) * 64 + col] = sum;
        }
        __syncthreads();
    }

    if (p.deps_partial) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);
        const double w = wave_sum_d(dot);
        if (lane == 0) red[wave] = w;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < nwaves; ++i) s += red[i];
            p.deps_partial[blockIdx.x] = s;
        }
    }
}

#ifdef GNM_AGG16_TUNING
static unsigned long long* g_agg16_stamps = nullptr;
extern "C" void gnm_debug_set_stamps(void* p) { g_agg16_stamps = reinterpret_cast<unsigned long long*>(p); }
#endif

static int launch_agg16(const AggArgs& a0, int B, int n_max, hipStream_t stream) {
    AggArgs a = a0;
#ifdef GNM_AGG16_TUNING
    a.stamps = g_agg16_stamps;
#else
    a.stamps = nullptr;
#endif
    size_t lds = (size_t)(n_max + 1) * 256 + (size_t)(n_max + 2) * 4 + 16;
    if (a.p_scale) lds += 16 + (size_t)1024 * 16;     // forward prologue: readout partials of up to 1024 threads
    GNM_ALLOW_FULL_LDS(&gnm_agg16_kernel<false>);
    GNM_ALLOW_FULL_LDS(&gnm_agg16_kernel<true>);
    int threads = 1024;
    if (lds <= 20 * 1024) threads = 256;
    else if (lds <= 48 * 1024) threads = 512;
    static const int env_threads = gnm_env_int("GNM_AGG16_THREADS", 0);   // tuning knob for tools/bench_agg.py
    if (env_threads >= 64 && env_threads <= 1024 && (env_threads & 63) == 0) threads = env_threads;
    if (reinterpret_cast<uintptr_t>(a.col) & 3) return GNM_ERR_UNSUPPORTED;     // column ids are fetched as aligned pairs
    // the reductions at the end of the kernel reuse the tile's LDS: [64] d-eps slots + [waves][2][64] doubles of column
    // statistics.  A batch of very small graphs (n_max < 17) has a tile smaller than that (found by
    // tests/test_gpu_aggm.py, graphs of 3 nodes: the sums lost every contribution written past the allocation)
    const size_t red = (size_t)(64 + (threads / 64) * 128) * 8;
    if (lds < red) lds = red;
    if (a.sZ)
        hipLaunchKernelGGL(gnm_agg16_kernel<true>, dim3(B * a.nslices), dim3(threads), lds, stream, a);
    else
        hipLaunchKernelGGL(gnm_agg16_kernel<false>, dim3(B * a.nslices), dim3(threads), lds, stream, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

template <int LPR>
static int launch_agg(const AggArgs& a0, int B, int n_max, hipStream_t stream) {
    AggArgs a = a0;
#ifdef GNM_AGG16_TUNING
    a.stamps = g_agg16_stamps;
#else
    a.stamps = nullptr;
#endif
    size_t lds = (size_t)(n_max + (LPR == 8 ? 2 : 1)) * LPR * 16;      // + the zero row(s)
    const int max_nnz = a.ids_in_lds;             // on entry: largest nnz of the batch (0 = unknown)
    a.ids_in_lds = 0;
    const int mode = LPR == 8 ? (a.sZ ? 2 : (a.p_scale ? 1 : 0)) : 0;      // fused forms (gnm_agg_kernel's MODE)
    if (LPR != 8 && (a.sZ || a.p_scale)) return GNM_ERR_UNSUPPORTED;
    const size_t extra = mode == 1 ? 16 * 8 * 16 : 0;                      // prologue: [16 waves][8] float4 of readout shares
    if (LPR == 4 && max_nnz > 0 && lds + (size_t)max_nnz * 2 + 96 <= (size_t)kLdsBudget - 1024) {
        a.ids_in_lds = 1;
        lds += (size_t)max_nnz * 2 + 96;          // + alignment shift and 16-B rounding of the staged id block
    } else {
        lds += (size_t)(n_max + 2) * 4 + 16;       // staged row offsets (gnm_agg_slice_width budgets for them)
        if (LPR == 8) {                             // + the waves' id scratch (16 x 1 KB), where it fits next to the tile
            const size_t with_scr = ((lds + 15) & ~(size_t)15) + 16 * 1024;
            if (with_scr + extra <= (size_t)kLdsBudget - 1024 && lds > 48 * 1024) {   // (1024-thread launches only: 16 waves)
                lds = with_scr;
                a.ids_in_lds = 2;
            }
        }
    }
    if (extra) {
        lds = ((lds + 15) & ~(size_t)15) + extra;
        if (lds > (size_t)kLdsBudget - 1024) return GNM_ERR_UNSUPPORTED;
    }
    GNM_ALLOW_FULL_LDS(&gnm_agg_kernel<LPR>);
    // one workgroup per CU when the tile is large (16 waves to keep the LDS pipe busy);
    // smaller tiles share a CU, so use fewer waves per workgroup.
    int threads = 1024;
    if (lds <= 20 * 1024) threads = 256;
    else if (lds <= 48 * 1024) threads = 512;
    if (a.ids_in_lds) {                            // one thread per (row, chunk): enough threads for a whole graph
        threads = ((n_max * LPR + 63) / 64) * 64;
        if (threads > 1024) threads = 1024;
        if (threads < 256) threads = 256;
    }
    if constexpr (LPR == 8) {
        if (mode) {
            // the STATS reductions at the end reuse the tile's first bytes: [64] d-eps slots + [waves][2][32] doubles
            const size_t red = (size_t)(64 + (threads / 64) * 64) * 8;
            if (lds < red) lds = red;
            if (mode == 1) {
                GNM_ALLOW_FULL_LDS((&gnm_agg_kernel<8, 1>));
                hipLaunchKernelGGL((gnm_agg_kernel<8, 1>), dim3(B * a.nslices), dim3(threads), lds, stream, a);
            } else {
                GNM_ALLOW_FULL_LDS((&gnm_agg_kernel<8, 2>));
                hipLaunchKernelGGL((gnm_agg_kernel<8, 2>), dim3(B * a.nslices), dim3(threads), lds, stream, a);
            }
            GNM_CHECK_LAUNCH();
            return GNM_OK;
        }
    }
    hipLaunchKernelGGL(gnm_agg_kernel<LPR>, dim3(B * a.nslices), dim3(threads), lds, stream, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---- graphs too large for an LDS-resident slice (round 3) ---------------------------------------------------------
// SURVEY.md 8(d), sparse row: "row-gather from L2/HBM, one wave per destination row".  The kernels above keep a graph's
// [n, FS] slice in LDS, which bounds n to ~4,500 nodes at the narrowest slice; beyond that (up to the 65,535 nodes the
// 16-bit column ids allow) this kernel gathers neighbour rows straight from global memory: a workgroup takes 64
// destination rows of one graph, a wave one row at a time, a lane one column of the current 64-column chunk (4-byte
// accesses: no alignment demands; a neighbour row's chunk is one 256-byte request).  Same semantics as gnm_agg_kernel
// (pre / post degree scaling, self loop or (1 + eps) self term, d-eps partial per workgroup in a fixed order); not
// tuned -- it exists so that no graph the arena can hold is refused.
static constexpr int kAggGRows = 64;
__global__ void __launch_bounds__(256) gnm_agg_global_kernel(const AggArgs p, int chunks) {
    __shared__ double red[4];
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int row0 = p.node_off[b];
    const int n = p.node_off[b + 1] - row0;
    const int32_t* rp = p.rowptr + p.b_rp_off[b];
    const uint16_t* cl = p.col + p.b_col_off[b];
    const int32_t* drp = p.deg_rowptr + p.b_deg_off[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool prescale = p.backward && p.average;
    const float selfB = p.self_loop ? 0.f : (p.eps ? 1.f + *p.eps : 1.f);
    double dot = 0.0;
    for (int v = ch * kAggGRows + wave; v < min(n, (ch + 1) * kAggGRows); v += 4) {
        const int e0 = rp[v], e1 = rp[v + 1];
        const float dv_in = (float)(drp[v + 1] - drp[v] + p.self_loop);      // what the pre-scale divides row v by
        for (int c0 = 0; c0 < p.F; c0 += 64) {
            const int c = c0 + lane;
            const bool on = c < p.F;
            float acc = 0.f;
            for (int e = e0; e < e1; ++e) {
                const int u = cl[e];
                float xv = on ? p.x[(size_t)(row0 + u) * p.ldx + c] : 0.f;
                if (prescale) {           // a row nobody's forward gathered has d = 0 and is never read here
                    const float du = (float)(drp[u + 1] - drp[u] + p.self_loop);
                    xv = xv / du;
                }
                acc += xv;
            }
            const float xs = on ? p.x[(size_t)(row0 + v) * p.ldx + c] : 0.f;
            const float own = prescale ? xs / dv_in : xs;
            if (p.deps_partial && on) dot += (double)xs * (double)p.hfwd[(size_t)(row0 + v) * p.ldh + c];
            if (p.y) {
                float tot = acc;
                if (p.self_loop) tot += own;
                if (p.average && !p.backward) tot /= (float)(e1 - e0 + p.self_loop);   // 0/0 -> NaN as the reference
                if (!p.self_loop) tot += selfB * xs;
                if (on) p.y[(size_t)(row0 + v) * p.ldy + c] = tot;
            }
        }
    }
    if (p.deps_partial) {
        const double w = wave_sum_d(dot);
        if (lane == 0) red[wave] = w;
        __syncthreads();
        if (threadIdx.x == 0) p.deps_partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

static int agg_global_chunks(int n_max) { return n_max > 0 ? (n_max + kAggGRows - 1) / kAggGRows : 1; }

static int launch_agg_global(const AggArgs& a, int B, int n_max, hipStream_t stream) {
    const int chunks = agg_global_chunks(n_max);
    hipLaunchKernelGGL(gnm_agg_global_kernel, dim3((unsigned)B * chunks), dim3(256), 0, stream, a, chunks);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// Feature-slice width (floats) the launcher will use for width F and the largest
// graph of the batch; 0 when even the narrowest slice does not fit in LDS.
extern "C" int gnm_agg_slice_width(int F, int n_max) {
    int fs = 8;
    static const int env_cap = gnm_env_int("GNM_AGG_SLICE_MAX", 128);   // tuning knob: widest slice tried
    while (fs < F && fs < 128 && fs < env_cap) fs <<= 1;
    while (fs >= 8 && (size_t)(n_max + 2) * fs * 4 + (size_t)(n_max + 2) * 4 + 16 > (size_t)kLdsBudget - 1024) fs >>= 1;
    return fs >= 8 ? fs : 0;
}

extern "C" int gnm_agg(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off,
                       const int64_t* b_col_off, const int32_t* deg_rowptr, const int64_t* b_deg_off,
                       const int32_t* node_off, int B, int n_max, int nnz_max, const float* x, int ldx, float* y, int ldy,
                       int F, const float* eps, int average, int self_loop, int backward, const float* hfwd,
                       int ldh, double* deps_partial, void* stream) {
    if (B <= 0) return GNM_OK;
    if (F <= 0 || n_max < 0 || n_max > 65535) return GNM_ERR_BAD_ARG;
    const int fs = gnm_agg_slice_width(F, n_max);
    AggArgs a;
    memset(&a, 0, sizeof(a));
    a.rowptr = rowptr; a.col = col; a.b_rp_off = b_rp_off; a.b_col_off = b_col_off;
    a.deg_rowptr = deg_rowptr ? deg_rowptr : rowptr;
    a.b_deg_off = b_deg_off ? b_deg_off : b_rp_off;
    a.node_off = node_off; a.x = x; a.y = y; a.eps = eps; a.hfwd = hfwd; a.deps_partial = deps_partial;
    a.ldx = ldx; a.ldy = ldy; a.ldh = ldh; a.F = F;
    a.nslices = fs > 0 ? (F + fs - 1) / fs : 1;
    a.average = average; a.self_loop = self_loop; a.backward = backward;
    a.debug = 0;
    a.ids_in_lds = nnz_max > 0 ? nnz_max : 0;   // launch_agg turns this into the 0/1 flag
    static const int env_debug = gnm_env_int("GNM_AGG16_DEBUG", 0);   // only acted on by -DGNM_AGG16_TUNING builds
    a.debug = env_debug;
    if (deps_partial && !hfwd) return GNM_ERR_BAD_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (fs == 0) {              // no slice of this graph fits in LDS: gather from global memory
        a.nslices = 1;
        return launch_agg_global(a, B, n_max, s);
    }
    switch (fs) {
        case 8: return launch_agg<2>(a, B, n_max, s);
        case 16: return launch_agg<4>(a, B, n_max, s);
        case 32: return launch_agg<8>(a, B, n_max, s);
        case 64: return launch_agg16(a, B, n_max, s);
        case 128: return launch_agg<32>(a, B, n_max, s);
    }
    return GNM_ERR_UNSUPPORTED;
}

// Aggregation backward fused with the BatchNorm-backward statistics of the layer below
// (see AggArgs): only for the 64-wide single-slice shape; GNM_ERR_UNSUPPORTED otherwise.
static AggArgs g_stats_none() {
    AggArgs z;
    memset(&z, 0, sizeof(z));
    return z;
}
extern "C" int gnm_agg_bwd_stats(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off,
                                 const int64_t* b_col_off, const int32_t* deg_rowptr, const int64_t* b_deg_off,
                                 const int32_t* node_off, int B, int n_max, int nnz_max, const float* x, int ldx,
                                 float* y, int ldy, int F, const float* eps, int average, int self_loop,
                                 const float* hfwd, int ldh, double* deps_partial, const float* sZ, int ldsz,
                                 const float* s_scale, const float* s_shift, const float* s_mean, const float* s_rstd,
                                 const float* dpool, int ld_dpool, int graph_avg, const float* dsc1, const float* U,
                                 int ld_U, const int32_t* inv_perm, const float* s2sum, double* s_partial,
                                 void* stream) {
    if (B <= 0) return GNM_OK;
    // two shapes: one 64-wide slice (gnm_agg16_kernel), or 32-float slices of a width that is a multiple of 32
    // (gnm_agg_kernel<8, 2>: hidden_dim 128 on graphs too large for a wider slice, configs[3])
    const int fs = gnm_agg_slice_width(F, n_max);
    const bool wide64 = F == 64 && fs == 64, sliced32 = fs == 32 && (F & 31) == 0;
    if ((!wide64 && !sliced32) || !y || !sZ || !s_partial) return GNM_ERR_UNSUPPORTED;
    if ((ldsz & 3) || (ldy & 3) || (ldx & 3) || (dpool && (ld_dpool & 3)) || (dsc1 && (ld_U & 3))) return GNM_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(x) & 15) return GNM_ERR_UNSUPPORTED;
    const uintptr_t al = reinterpret_cast<uintptr_t>(sZ) | reinterpret_cast<uintptr_t>(s_scale) |
                         reinterpret_cast<uintptr_t>(s_shift) | reinterpret_cast<uintptr_t>(s_mean) |
                         reinterpret_cast<uintptr_t>(s_rstd) | reinterpret_cast<uintptr_t>(dpool) |
                         reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(y);
    if (al & 15) return GNM_ERR_UNSUPPORTED;
    // deps_partial with hfwd == NULL: h is recomputed from sZ in the epilogue (needs the (1+eps) self-term form)
    if (deps_partial && !hfwd && self_loop) return GNM_ERR_BAD_ARG;
    AggArgs a = g_stats_none();
    a.rowptr = rowptr; a.col = col; a.b_rp_off = b_rp_off; a.b_col_off = b_col_off;
    a.deg_rowptr = deg_rowptr ? deg_rowptr : rowptr;
    a.b_deg_off = b_deg_off ? b_deg_off : b_rp_off;
    a.node_off = node_off; a.x = x; a.y = y; a.eps = eps; a.hfwd = hfwd; a.deps_partial = deps_partial;
    a.ldx = ldx; a.ldy = ldy; a.ldh = ldh; a.F = F; a.nslices = 1;
    a.average = average; a.self_loop = self_loop; a.backward = 1;
    a.sZ = sZ; a.s_scale = s_scale; a.s_shift = s_shift; a.s_mean = s_mean; a.s_rstd = s_rstd;
    a.s_dpool = dpool; a.s_dsc1 = dsc1; a.s_U = U; a.s_inv_perm = inv_perm; a.s_s2sum = s2sum;
    a.s_partial = s_partial; a.ldsz = ldsz; a.ld_dpool = ld_dpool; a.ld_U = ld_U; a.s_avg = graph_avg;
    a.n_batch = B;
    if (sliced32) {
        a.nslices = F / 32;
        a.ids_in_lds = nnz_max > 0 ? nnz_max : 0;
        return launch_agg<8>(a, B, n_max, reinterpret_cast<hipStream_t>(stream));
    }
    return launch_agg16(a, B, n_max, reinterpret_cast<hipStream_t>(stream));
}

// Forward aggregation of layer l+1 whose tile load is layer l's outer BatchNorm + ReLU + graph readout
// (graphcnn.py:163-166, 229 folded into :154-161): x is Z of layer l's last Linear; hout receives
// h_l = relu(Z * scale + shift), gf[b, :] its sum (mean) over graph b's nodes, y the aggregation of h_l.
// Only the 64-wide single-slice LDS shape; GNM_ERR_UNSUPPORTED otherwise (caller: gnm_bn_relu_readout + gnm_agg).
extern "C" int gnm_agg_fwd_bnrelu(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off,
                                  const int64_t* b_col_off, const int32_t* node_off, int B, int n_max, int nnz_max,
                                  const float* z, int ldz, const float* scale, const float* shift, float* hout,
                                  int ldh, float* gf, int ldgf, int graph_avg, float* y, int ldy, int F,
                                  const float* eps, int average, int self_loop, void* stream) {
    if (B <= 0) return GNM_OK;
    // two shapes, as gnm_agg_bwd_stats: one 64-wide slice, or 32-float slices of a multiple of 32 (gnm_agg_kernel<8, 1>)
    const int fs = gnm_agg_slice_width(F, n_max);
    const bool wide64 = F == 64 && fs == 64, sliced32 = fs == 32 && (F & 31) == 0;
    if ((!wide64 && !sliced32) || !y || !z || !scale || !shift) return GNM_ERR_UNSUPPORTED;
    if (wide64 && (size_t)(n_max + 1) * 256 + (size_t)(n_max + 2) * 4 + 32 + (size_t)1024 * 16 > (size_t)kLdsBudget - 1024)
        return GNM_ERR_UNSUPPORTED;
    if ((ldz & 3) || (hout && (ldh & 3)) || (ldy & 3) || (gf && (ldgf & 3))) return GNM_ERR_UNSUPPORTED;
    const uintptr_t al = reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(scale) |
                         reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(hout) |
                         reinterpret_cast<uintptr_t>(gf) | reinterpret_cast<uintptr_t>(y);
    if (al & 15) return GNM_ERR_UNSUPPORTED;
    AggArgs a = g_stats_none();
    a.rowptr = rowptr; a.col = col; a.b_rp_off = b_rp_off; a.b_col_off = b_col_off;
    a.deg_rowptr = rowptr; a.b_deg_off = b_rp_off;
    a.node_off = node_off; a.x = z; a.y = y; a.eps = eps;
    a.ldx = ldz; a.ldy = ldy; a.F = F; a.nslices = 1;
    a.average = average; a.self_loop = self_loop; a.backward = 0;
    a.p_scale = scale; a.p_shift = shift; a.p_hout = hout; a.p_gf = gf; a.p_ldh = ldh; a.p_ldgf = ldgf;
    a.p_gf_avg = graph_avg;
    if (sliced32) {
        a.nslices = F / 32;
        a.ids_in_lds = nnz_max > 0 ? nnz_max : 0;
        return launch_agg<8>(a, B, n_max, reinterpret_cast<hipStream_t>(stream));      // (declines when LDS has no room for the shares)
    }
    return launch_agg16(a, B, n_max, reinterpret_cast<hipStream_t>(stream));
}

// Number of deps partials gnm_agg writes for (F, n_max, B): B * nslices.
extern "C" int gnm_agg_num_partials(int F, int n_max, int B) {
    const int fs = gnm_agg_slice_width(F, n_max);
    if (fs == 0) return B * agg_global_chunks(n_max);      // the global-gather kernel: one partial per workgroup
    return B * ((F + fs - 1) / fs);
}
