// GIN neighbour aggregation on the matrix cores (gfx950), for graphs dense enough that gathering pays less than
// multiplying: same operation, same fused prologues / epilogues and same C-ABI argument meaning as agg.hip
// (torch.spmm(Adj_block, h) + degree / (1 + eps) terms, /root/reference models/graphcnn.py:154-161, :178-182 and its
// autograd backward), but the graph's adjacency is a BIT matrix and the neighbour sum is  A (0/1) x h  on MFMA.
//
// Why: the benchmark graphs (400 nodes, mean degree 119: 30 % dense) make the LDS gather of agg.hip read 119 x 256 B
// per output row; at ~86 % of the LDS peak that is ~160 us per layer launch and cannot go lower.  As a product the
// same sum is 13 x 2 x 25 MFMA 32x32x16 steps per graph and operand plane; the bit matrix is 21 KB per graph instead
// of 95 KB of column ids, and HBM (features in, result out) becomes the bound.
//
// fp32 exactness: the matrix cores multiply bf16.  A is 0/1 (exact).  h is split into three bf16 planes by
// truncation, h = h1 + h2 + h3 with h1 = top 16 bits of h, h2 = top 16 bits of (h - h1), h3 = h - h1 - h2 (8
// significant bits each, every subtraction exact), so every product is exact and the only rounding is the fp32
// accumulation inside the MFMA -- the same kind and size of error as the fp32 sum of the gather (whose order is not
// the reference's either).  tests/test_gpu_aggm.py holds it to the CSR kernels and the fp64 oracle.
// One difference by construction: a non-finite feature (inf / NaN) reaches every row of its graph (0 x NaN = NaN),
// not only its neighbours as in the gather -- either way the forward is lost.
//
// One workgroup (512 threads, two per CU) = one graph x one 32-column block of the feature matrix:
//   phase A  the [n, 32] tile is read from HBM once (16 B per lane), the prologue of the launch form is applied
//            (BatchNorm + ReLU + readout, or the 1/deg pre-scale of the "average" backward), and the three planes
//            are written to LDS transposed into the MFMA B-operand order [k / 8][column][k % 8];
//   phase B  wave w owns output row blocks w and w + 8 (32 rows each).  Per 16-node step it expands its rows' 16
//            adjacency bits into the bf16 A operand through a 16-entry LDS table (nibble -> four bf16), reads the three
//            B fragments (ds_read_b128, lane-linear: conflict free) and issues 3 (6) MFMAs;
//   epilogue on the accumulators (lane = column, 16 rows per lane): self term, degree division, the fused
//            backward terms and BatchNorm statistics of gnm_agg_bwd_stats, 128-B row segments stored.
// The row-block -> wave map is static, so every reduction (column statistics, d eps, readout) has a fixed order.
#include "gnm_agg_args.h"
#include <string.h>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifndef GNM_AGGM_WAVES
#define GNM_AGGM_WAVES 8
#endif
static constexpr int kAggmWaves = GNM_AGGM_WAVES;   // row blocks w and w + kAggmWaves per wave (13 blocks at n = 400)
static constexpr int kAggmThreads = 64 * kAggmWaves;
static constexpr int kAggmMaxN = 416;            // 13 row blocks; up to 400 nodes two workgroups share a CU's LDS
static constexpr int kAggmScratch = 128 + 1024;  // nibble table + readout partials
// plane layout: [k / 8][32 columns][8 consecutive k] bf16, 528 bytes per k-group (512 + 16 of padding: the 8-byte
// transposing writes of phase A then spread over all banks; the 16-byte reads of phase B are lane-linear either way)
static constexpr unsigned kAggmK8Stride = 528;
static constexpr unsigned kAggmStepBytes = 2 * kAggmK8Stride;    // one 16-node MFMA step = two k-groups

#ifdef GNM_AGG16_TUNING       // in-kernel timeline (tools/aggm_timeline.py)
static unsigned long long* g_aggm_stamps = nullptr;
extern "C" void gnm_debug_set_aggm_stamps(void* p) { g_aggm_stamps = reinterpret_cast<unsigned long long*>(p); }
#define GNM_MSTAMP(k)                                                                                         \
    if (p.stamps && (threadIdx.x & 63) == 0)                                                                  \
        p.stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime();   /* <= 8 waves */
#else
#define GNM_MSTAMP(k)
#endif

// words per HALF row of the bit adjacency (layout: see gnm_adj_bits_build below)
__host__ __device__ static inline int aggm_half_words(int W) { return (((W + 1) >> 1) + 3) & ~3; }

__device__ __forceinline__ unsigned bf16_pair_hi(unsigned lo_word, unsigned hi_word) {
    // (top 16 bits of hi_word) : (top 16 bits of lo_word)
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);
}

// AVG: neighbor_pooling_type "average" (the epilogue then also loads degrees / the raw input); decided by the launcher
// so that the "sum" forms carry none of it.
// NARROW: F < 32 (the input layer, F0 = 7 in the benchmark): one partial column block, rows of x not 16-byte
// addressable -- 4-byte loads with column guards, zero planes past F, stores only for columns < F.  Plain launch form
// only (no fused prologue, no statistics, no d-eps).
template <bool STATS, bool AVG, bool NARROW = false>
__global__ void __launch_bounds__(kAggmThreads, 4) gnm_aggm_kernel(const AggArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- which graph / column block: the blocks of one graph sit 8 apart, i.e. on the same XCD (one L2 serves the
    //      bit matrix and the tile to all of them)
    const int nc = NARROW ? 1 : p.F >> 5;                         // 32-column blocks per graph
    const int grp = blockIdx.x / (8 * nc), within = blockIdx.x - grp * (8 * nc);
    const int b = grp * 8 + (within & 7);
    const int cb = within >> 3;
    if (b >= p.n_graphs) return;
    const int row0 = p.node_off[b];
    const int n = p.node_off[b + 1] - row0;
    const int col0 = cb * 32;
    if (n <= 0) {            // an empty graph: no rows to write, its reductions are empty sums
        const int t = threadIdx.x;
        if (p.p_scale && p.p_gf && t < 32) p.p_gf[(size_t)b * p.p_ldgf + col0 + t] = p.p_gf_avg ? 0.f / 0.f : 0.f;
        if (STATS && t < 64) p.s_partial[((size_t)b * 2 + (t >> 5)) * 64 + col0 + (t & 31)] = 0.0;
        if (p.deps_partial && t == 0) p.deps_partial[(size_t)b * nc + cb] = 0.0;
        return;
    }
    const int W = (n + 31) >> 5;                                 // words per bit row = 32-row blocks
    const int ksteps = (n + 15) >> 4;
    const int n16 = ksteps * 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const unsigned plane_bytes = (unsigned)(p.n16_max >> 3) * kAggmK8Stride;
    char* lut = smem + 3u * plane_bytes;
    float4* rsum = reinterpret_cast<float4*>(lut + 128);
    const bool prescale = AVG && p.backward;
    const bool pro = !STATS && p.p_scale != nullptr;
    const bool dot_a = p.deps_partial && p.hfwd;
    const int32_t* drp = p.deg_rowptr + p.b_deg_off[b];

    GNM_MSTAMP(0)
    // This wave's output row blocks and their adjacency bits: requested first, so they arrive under phase A (the
    // timeline of the first version showed every wave waiting ~20 % of the workgroup's life for them behind the barrier)
    const int role = (wave + cb) % kAggmWaves;                  // rotate with the column block: evens out the SIMDs
    const int rbA = role, rbB = role + kAggmWaves;
    const bool two = rbB < W;
    const bool has_rows = p.y && rbA < W;
    // Bit rows (round 3 layout, see gnm_adj_bits_build): a row's bytes are stored de-interleaved -- the even bytes
    // (columns 16 s .. 16 s + 7 of MFMA step s: what lanes 0-31 multiply) in the first half of the row, the odd bytes
    // (columns 16 s + 8 .. + 15: lanes 32-63) in the second, each half padded to 16-byte pieces.  A lane therefore
    // loads exactly the bytes it uses, byte m of word j = step 4 j + m, as one or two 16-byte pieces: no permutes
    // and half the registers of the interleaved layout (which every lane had to load whole).
    const int HPW = (((W + 1) >> 1) + 3) & ~3;                   // words per half row (4 or 8)
    unsigned pkA[8], pkB[8];
    {
        const uint32_t* gbits = p.adj_bits + p.b_bits_off[b];
        const u32x4* ra = reinterpret_cast<const u32x4*>(gbits + (size_t)(min(rbA, W - 1) * 32 + i) * (2 * HPW) + h * HPW);
        const u32x4* rb = reinterpret_cast<const u32x4*>(gbits + (size_t)((two ? rbB : min(rbA, W - 1)) * 32 + i) * (2 * HPW) + h * HPW);
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        const u32x4 a0 = ra[0], b0 = rb[0];
        const u32x4 a1 = HPW > 4 ? ra[1] : z4, b1 = HPW > 4 ? rb[1] : z4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pkA[j] = a0[j]; pkA[4 + j] = a1[j];
            pkB[j] = b0[j]; pkB[4 + j] = b1[j];
        }
    }
    if (tid < 16) {          // nibble e -> bf16 (bit 0, bit 1, bit 2, bit 3) as two words
        const unsigned one = 0x3F80u;
        u32x2 v;
        v.x = ((tid & 1) ? one : 0u) | ((tid & 2) ? one << 16 : 0u);
        v.y = ((tid & 4) ? one : 0u) | ((tid & 8) ? one << 16 : 0u);
        *reinterpret_cast<u32x2*>(lut + 8 * tid) = v;
    }

    // ---- phase A ------------------------------------------------------------------------------
    // item = (4 consecutive rows, 4 consecutive columns): four 16-B loads, transposed in registers into 8-B pieces
    // of the planes (4 consecutive k of one column)
    const int c4 = tid & 7;
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pro) {
        psc = *reinterpret_cast<const float4*>(p.p_scale + col0 + 4 * c4);
        psh = *reinterpret_cast<const float4*>(p.p_shift + col0 + 4 * c4);
    }
    double dot = 0.0;
    const int nitems = (n16 >> 2) * 8;
    constexpr int UA = (832 + kAggmThreads - 1) / kAggmThreads;   // 416 / 4 * 8 = 832 items
    float4 v[UA][4];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int it = tid + u * kAggmThreads;
        const int rq = it >> 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {     // unconditional (clamped to the graph's last row; masked below): no branches
            const float* src = p.x + (size_t)(row0 + min(4 * rq + r, n - 1)) * p.ldx + col0 + 4 * c4;
            if constexpr (!NARROW) {
                v[u][r] = *reinterpret_cast<const float4*>(src);
            } else {                      // columns past F: re-read the row's last column, zeroed below
                const int cc = 4 * c4, last = p.F - 1;
                const float* rowp = src - cc;
                v[u][r] = make_float4(rowp[min(cc, last)], rowp[min(cc + 1, last)], rowp[min(cc + 2, last)], rowp[min(cc + 3, last)]);
            }
        }
    }
    GNM_MSTAMP(1)
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int it = tid + u * kAggmThreads;
        const int rq = it >> 3;
        if (it < nitems) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * rq + r;
                float4 w = row < n ? v[u][r] : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (NARROW) {
                    const int cc = 4 * c4;
                    if (cc + 0 >= p.F) w.x = 0.f;
                    if (cc + 1 >= p.F) w.y = 0.f;
                    if (cc + 2 >= p.F) w.z = 0.f;
                    if (cc + 3 >= p.F) w.w = 0.f;
                }
                if (row < n) {
                    if (dot_a) {
                        const float4 hh = *reinterpret_cast<const float4*>(p.hfwd + (size_t)(row0 + row) * p.ldh + col0 + 4 * c4);
                        dot += (double)w.x * hh.x + (double)w.y * hh.y + (double)w.z * hh.z + (double)w.w * hh.w;
                    }
                    if (pro) {
                        w.x = gnm_relu(w.x * psc.x + psh.x); w.y = gnm_relu(w.y * psc.y + psh.y);
                        w.z = gnm_relu(w.z * psc.z + psh.z); w.w = gnm_relu(w.w * psc.w + psh.w);
                        if (p.p_hout) *reinterpret_cast<float4*>(p.p_hout + (size_t)(row0 + row) * p.p_ldh + col0 + 4 * c4) = w;
                        csum.x += w.x; csum.y += w.y; csum.z += w.z; csum.w += w.w;
                    }
                    if (prescale) {
                        // d == 0: a row nobody gathers (no forward neighbours, no self loop).  The gather never reads
                        // its x / 0; a product would multiply it by a zero bit (0 x inf = NaN): keep it out
                        const float d = (float)(drp[row + 1] - drp[row] + p.self_loop);
                        const float inv_ok = d > 0.f ? 1.f : 0.f;
                        w.x = inv_ok != 0.f ? w.x / d : 0.f; w.y = inv_ok != 0.f ? w.y / d : 0.f;
                        w.z = inv_ok != 0.f ? w.z / d : 0.f; w.w = inv_ok != 0.f ? w.w / d : 0.f;
                    }
                }
                v[u][r] = w;
            }
            // three planes by truncation; element (row 4 rq + r, column 4 c4 + c) -> plane word index below
            const unsigned base = (unsigned)(rq >> 1) * kAggmK8Stride + (unsigned)((4 * c4 * 8 + 4 * (rq & 1)) * 2);   // bytes, column 4 c4
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned x0[4], x1[4], x2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float f = c == 0 ? v[u][r].x : (c == 1 ? v[u][r].y : (c == 2 ? v[u][r].z : v[u][r].w));
                    const unsigned a1 = __float_as_uint(f) & 0xFFFF0000u;
                    const float r1 = f - __uint_as_float(a1);
                    const unsigned a2 = __float_as_uint(r1) & 0xFFFF0000u;
                    const float r2 = r1 - __uint_as_float(a2);
                    x0[r] = a1; x1[r] = a2; x2[r] = __float_as_uint(r2);
                }
                u32x2 w0, w1, w2;
                w0.x = bf16_pair_hi(x0[0], x0[1]); w0.y = bf16_pair_hi(x0[2], x0[3]);
                w1.x = bf16_pair_hi(x1[0], x1[1]); w1.y = bf16_pair_hi(x1[2], x1[3]);
                w2.x = bf16_pair_hi(x2[0], x2[1]); w2.y = bf16_pair_hi(x2[2], x2[3]);
                char* dst = smem + base + c * 16;
                *reinterpret_cast<u32x2*>(dst) = w0;
                *reinterpret_cast<u32x2*>(dst + plane_bytes) = w1;
                *reinterpret_cast<u32x2*>(dst + 2u * plane_bytes) = w2;
            }
        }
    }
    if (pro && p.p_gf) {       // readout partials: lanes with the same column chunk (lane & 7), then the waves
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            csum.x += __shfl_xor(csum.x, off, 64); csum.y += __shfl_xor(csum.y, off, 64);
            csum.z += __shfl_xor(csum.z, off, 64); csum.w += __shfl_xor(csum.w, off, 64);
        }
        if (lane < 8) rsum[wave * 8 + lane] = csum;
    }
    GNM_MSTAMP(2)
    __syncthreads();
    if (pro && p.p_gf && tid < 8) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < kAggmThreads / 64; ++w) {
            const float4 s = rsum[w * 8 + tid];
            t.x += s.x; t.y += s.y; t.z += s.z; t.w += s.w;
        }
        if (p.p_gf_avg) {
            const float inv = 1.f / (float)n;
            t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
        }
        *reinterpret_cast<float4*>(p.p_gf + (size_t)b * p.p_ldgf + col0 + 4 * tid) = t;
    }

    // ---- phase B ------------------------------------------------------------------------------
    const float selfB = p.self_loop ? 0.f : (p.eps ? 1.f + *p.eps : 1.f);
    const bool need_deg = AVG && !p.backward;
    const int col = col0 + i;
    float lsc = 0.f, lsh = 0.f, lmu = 0.f, s_pb = 0.f, s_ub = 0.f;
    float ss1 = 0.f, ss2 = 0.f;
    if constexpr (STATS) {
        lsc = p.s_scale[col]; lsh = p.s_shift[col]; lmu = p.s_mean[col];
        if (p.s_dpool) {
            s_pb = p.s_dpool[(size_t)b * p.ld_dpool + col];
            if (p.s_avg) s_pb *= 1.0f / (float)n;
        }
        if (p.s_dsc1) s_ub = p.s_U[(size_t)b * p.ld_U + col];
    }
    const char* bp0 = smem + h * kAggmK8Stride + i * 16;
    const char* bp1 = bp0 + plane_bytes;
    const char* bp2 = bp1 + plane_bytes;

    if (has_rows) {
        f32x16 accA, accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
        auto afrag = [&](unsigned pk, int m) -> bf16x8 {
            const unsigned byte3 = m == 0 ? (pk << 3) : (pk >> (8 * m - 3));
            const unsigned lo = byte3 & 0x78u, hi = (byte3 >> 4) & 0x78u;
            const u32x2 l2 = *reinterpret_cast<const u32x2*>(lut + lo);
            const u32x2 h2 = *reinterpret_cast<const u32x2*>(lut + hi);
            const u32x4 q = {l2.x, l2.y, h2.x, h2.y};
            return __builtin_bit_cast(bf16x8, q);
        };
        auto bfrag = [&](const char* bp, int ks) -> bf16x8 {
            return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + ks * kAggmStepBytes));
        };
        // The product, software-pipelined by hand: the operands of step ks + 1 are requested from LDS before the MFMAs
        // of step ks are issued (left to the compiler, every step was  ds_read x5 -> wait -> 3 MFMA -> ds_read x2 -> wait
        // -> 3 MFMA: two LDS round trips per 192 cycles of matrix work).  The MFMAs of the two row blocks alternate, so
        // no MFMA waits for the one issued right before it.  Steps past the graph's last are skipped by a wave-uniform
        // branch; what was requested for them is read from inside the workgroup's own LDS and dropped.
        auto product = [&](auto two_tag) {
            constexpr bool TWO = decltype(two_tag)::value;
            bf16x8 b0 = bfrag(bp0, 0), b1 = bfrag(bp1, 0), b2 = bfrag(bp2, 0);
            bf16x8 aA = afrag(pkA[0], 0), aB = aA;
            if constexpr (TWO) aB = afrag(pkB[0], 0);
#pragma unroll
            for (int ks = 0; ks < 26; ++ks) {
                if (ks < ksteps) {                                // wave-uniform
                    constexpr int LASTK = 25;
                    const int kn = ks < LASTK ? ks + 1 : LASTK;
                    const bf16x8 n0 = bfrag(bp0, kn), n1 = bfrag(bp1, kn), n2 = bfrag(bp2, kn);
                    const bf16x8 nA = afrag(pkA[kn >> 2], kn & 3);
                    bf16x8 nB = nA;
                    if constexpr (TWO) nB = afrag(pkB[kn >> 2], kn & 3);
                    __builtin_amdgcn_sched_barrier(0);            // the requests above stay above the MFMAs below
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b0, accA, 0, 0, 0);
                    if constexpr (TWO) accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aB, b0, accB, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b1, accA, 0, 0, 0);
                    if constexpr (TWO) accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aB, b1, accB, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b2, accA, 0, 0, 0);
                    if constexpr (TWO) accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aB, b2, accB, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    b0 = n0; b1 = n1; b2 = n2; aA = nA; aB = nB;
                }
            }
        };
        // ---- epilogue on the accumulators: lane = column, 16 rows per lane, in half blocks of 8 rows ------------
        // gfx950 retires vector-memory operations in issue order, stores included.  The first version asked for the
        // next 8 rows' operands after storing the previous 8 and so paid an HBM write round trip per half block (the
        // in-kernel timeline had the epilogues at 35-45 % of a workgroup's life -- even in the forward forms, which
        // load nothing: a conditional load the launch does not take still leaves its vmcnt(0) behind).  Here nothing
        // is conditional: rows past n are clipped by the store's buffer descriptor, an absent operand is loaded from
        // a valid address and discarded, and the requests for half block k + 1 are issued BEFORE the stores of half
        // block k, so the compiler's counted wait for them leaves those stores in flight.  The "sum" forward forms
        // load nothing and wait for nothing.
        const int32_t* frp = p.rowptr + p.b_rp_off[b];
        const bool need_xs = prescale && !p.self_loop;      // the (1 + eps) self term of the "average" backward: raw input
        const bool shuffled = STATS && p.s_dsc1 && row0 < p.n_batch;   // rows perm[g] < B of the shuffled branch (graphcnn.py:242)
        const bool has_dsc = STATS && p.s_dsc1 != nullptr;
        const float* dscp = has_dsc ? p.s_dsc1 : p.x;       // absent: any readable address, value discarded
        const bool want_dot = STATS && p.deps_partial && !p.hfwd && !p.self_loop;
        const unsigned ybytes = (unsigned)(((size_t)(n - 1) * p.ldy + p.F) * 4);
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)row0 * p.ldy, 0, (int)ybytes, 0x00020000);
        // quarter blocks: 4 rows per lane at a time (r = 4 k + q: rows rb * 32 + 8 k + 4 h + q); two quarters' operands
        // are in flight (with 8-row halves the two operand sets spilled at 128 registers)
        struct Ops { float zr[4], dv[4], xs[4]; int d0[4], d1[4]; };
        auto vrow_of = [&](int rb, int k, int q) { return rb * 32 + 8 * k + 4 * h + q; };
        auto request = [&](int rb, int k, Ops& o) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int vc = min(vrow_of(rb, k, q), n - 1);
                if constexpr (STATS) {
                    o.zr[q] = p.sZ[(size_t)(row0 + vc) * p.ldsz + col];
                    o.dv[q] = dscp[row0 + vc];
                }
                if constexpr (AVG) {
                    o.xs[q] = p.x[(size_t)(row0 + vc) * p.ldx + (NARROW ? min(col, p.F - 1) : col)];
                    o.d0[q] = frp[vc]; o.d1[q] = frp[vc + 1];
                }
            }
        };
        auto finish = [&](int rb, int k, const f32x16& acc, const Ops& o) {
            float ex[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) ex[q] = 0.f;
            if constexpr (STATS) {
                if (shuffled) {        // workgroup-uniform and rare (the workgroups of the first B rows of the batch)
                    int gq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) gq[q] = min(max(p.s_inv_perm[min(row0 + min(vrow_of(rb, k, q), n - 1), p.n_batch - 1)], 0), p.n_batch - 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = p.s_s2sum[gq[q]] * p.s_U[(size_t)gq[q] * p.ld_U + col];
                        ex[q] = row0 + min(vrow_of(rb, k, q), n - 1) < p.n_batch ? t : 0.f;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int vrow = vrow_of(rb, k, q);
                float tot = acc[4 * k + q];
                // the tile's own value of this element, as phase A formed it: the three planes add up to it exactly
                const char* e = smem + (unsigned)(min(vrow, n16 - 1) >> 3) * kAggmK8Stride + i * 16 + (vrow & 7) * 2;
                const float e1 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e) << 16);
                const float e2 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e + plane_bytes) << 16);
                const float e3 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e + 2u * plane_bytes) << 16);
                const float wv = (e1 + e2) + e3;
                if (p.self_loop) tot += wv;
                if constexpr (AVG) {
                    if (need_deg) tot /= (float)(o.d1[q] - o.d0[q] + p.self_loop);      // 0/0 -> NaN as in the reference
                }
                float sb = wv;
                if constexpr (AVG) sb = need_xs ? o.xs[q] : wv;
                if (!p.self_loop) tot += selfB * sb;
                if constexpr (STATS) {
                    const float zrow = o.zr[q];
                    if (want_dot && vrow < n) dot += (double)(sb * gnm_relu(zrow * lsc + lsh));   // h as the forward formed it
                    tot += s_pb + (has_dsc ? o.dv[q] : 0.f) * s_ub;
                    tot += ex[q];
                    if (!(zrow * lsc + lsh > 0.f)) tot = 0.f;
                    if (vrow < n) {
                        ss1 += tot;
                        ss2 += tot * (zrow - lmu);
                    }
                }
                // (row offset in the vector operand, scalar offset 0: see linear.hip, gnm_lin_stream_kernel)
                // (NARROW: a lane whose column is past F stores to an offset the descriptor clips)
                const unsigned yoff = (NARROW && col >= p.F) ? 0xFFFFFFF0u : (unsigned)((vrow * p.ldy + col) * 4);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tot), ry, yoff, 0, 0);
            }
        };
        Ops oa, ob;
        request(rbA, 0, oa);       // the first quarter's operands travel under the product
        GNM_MSTAMP(3)
        if (two) product(std::true_type{});
        else product(std::false_type{});
        GNM_MSTAMP(4)
        request(rbA, 1, ob);
        finish(rbA, 0, accA, oa);
        request(rbA, 2, oa);
        finish(rbA, 1, accA, ob);
        request(rbA, 3, ob);
        finish(rbA, 2, accA, oa);
        if (two) {
            request(rbB, 0, oa);
            finish(rbA, 3, accA, ob);
            GNM_MSTAMP(5)
            request(rbB, 1, ob);
            finish(rbB, 0, accB, oa);
            request(rbB, 2, oa);
            finish(rbB, 1, accB, ob);
            request(rbB, 3, ob);
            finish(rbB, 2, accB, oa);
            finish(rbB, 3, accB, ob);
        } else {
            finish(rbA, 3, accA, ob);
            GNM_MSTAMP(5)
        }
    }
    GNM_MSTAMP(6)

    // ---- reductions (fixed order) ---------------------------------------------------------------
    if constexpr (STATS) {
        __syncthreads();                                          // the planes are dead: reuse them
        double* sred = reinterpret_cast<double*>(smem);          // [8 waves][2][32]
        double d1 = (double)ss1, d2 = (double)ss2;
        d1 += __shfl_xor(d1, 32, 64);
        d2 += __shfl_xor(d2, 32, 64);
        if (h == 0) {
            sred[(wave * 2 + 0) * 32 + i] = d1;
            sred[(wave * 2 + 1) * 32 + i] = d2;
        }
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5, c = tid & 31;
            double sum = 0.0;
            for (int w = 0; w < kAggmThreads / 64; ++w) sum += sred[(w * 2 + which) * 32 + c];
            if (which) sum *= (double)p.s_rstd[col0 + c];         // sum G (Z - mean) -> sum G xhat
            p.s_partial[((size_t)b * 2 + which) * 64 + col0 + c] = sum;
        }
    }
    if (p.deps_partial) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);
        const double w = wave_sum_d(dot);
        if (lane == 0) red[wave] = w;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int k = 0; k < kAggmThreads / 64; ++k) s += red[k];
            p.deps_partial[(size_t)b * nc + cb] = s;
        }
    }
}

// ---- persistent form (round 3) ---------------------------------------------------------------------
// The kernel above starts a workgroup per (graph, column block): it loads its tile, waits for it (a full HBM round trip:
// 31 % of its life in the in-kernel timeline, profiles/r02_aggm_timeline.md), multiplies, stores, ends -- and because
// the ~512 resident workgroups of the chip do this more or less together, HBM sees bursts with idle time between them
// (0.34 of peak on the bytes moved).  Here ONE workgroup of 13 waves per CU stays resident and walks its units (graph x
// 32-column block) with TWO plane buffers in LDS.  In every unit a wave
//   1. requests its share of the NEXT unit's tile (one 4 x 4 item per thread: 16 registers in flight) and its rows
//      of the next unit's bit adjacency,
//   2. multiplies its 32-row block of the CURRENT unit (the same MFMA product as above) and runs the epilogue,
//   3. applies the launch form's prologue to the item that has meanwhile arrived (BatchNorm + ReLU + readout, 1/deg
//      pre-scale, d-eps dot), splits it into the three bf16 planes and writes them into the OTHER buffer,
// and one s_barrier hands the buffers over.  HBM reads travel under the products, stores drain under the next one, and
// no wave ever waits for a tile it has just asked for.  (A first version with dedicated loader waves -- 3 of 16 -- was
// loader-bound: 98 / 107 / 146 us against 85 / 86 / 118 for the per-unit kernel.)  Per-unit reductions (readout,
// BatchNorm-backward sums, d eps) go through parity-indexed LDS scratch and are finished after the hand-over by one
// wave, in a fixed order: results do not depend on timing.
// Needs both plane buffers in LDS: n_max <= 400 (two 75 KB buffers, unpadded 512-byte k-groups), F a multiple of 32.
static constexpr int kPWaves = 13;                    // = row blocks of a 400-node graph
static constexpr int kPThreads = 64 * kPWaves;        // 832 >= 800 items of a 400-row tile
static constexpr int kPMaxN = 400;
static constexpr unsigned kPK8 = 512;                 // bytes per k-group (8 rows x 32 columns bf16), unpadded
static constexpr unsigned kPStep = 2 * kPK8;
// scratch behind the plane buffers: nibble table | [2 parities] x 13 waves x 256 bytes: readout partials (8 float4 used)
// in the forward forms, column sums [2][32] float in the statistics form (the two never coexist) | d-eps partials
// [2 sides][2 parities][16] double
static constexpr unsigned kPWaveScratch = 256;
static constexpr unsigned kPScratch = 128 + 2 * kPWaves * kPWaveScratch + 2 * 2 * 16 * 8;

#ifdef GNM_AGG16_TUNING       // in-kernel timeline (tools/aggp_timeline.py): [workgroup][16 waves][8 units][8 stamps]
#define GNM_PSTAMP(it_, k_)                                                                                   \
    if (p.stamps && (it_) < 8 && (threadIdx.x & 63) == 0)                                                     \
        p.stamps[(((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (it_)) * 8 + (k_)] = __builtin_amdgcn_s_memtime();
#else
#define GNM_PSTAMP(it_, k_)
#endif

template <bool STATS, bool AVG>
__global__ void __launch_bounds__(kPThreads, 4) gnm_aggp_kernel(const AggArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int nc = p.F >> 5;
    const int total = ((p.n_graphs + 7) >> 3) * 8 * nc;          // units, padded to whole XCD rounds
    const unsigned plane_bytes = (unsigned)(p.n16_max >> 3) * kPK8;
    const unsigned buf_bytes = 3u * plane_bytes;
    char* lut = smem + 2u * buf_bytes;
    char* wscr = lut + 128;                                                    // [2][kPWaves][256 B]
    double* dred = reinterpret_cast<double*>(wscr + 2 * kPWaves * kPWaveScratch);   // [2 sides][2][16]
    const bool prescale = AVG && p.backward;
    const bool pro = !STATS && p.p_scale != nullptr;
    const bool dot_a = p.deps_partial && p.hfwd;                  // d eps against a given forward input: split side
    const bool want_dot = STATS && p.deps_partial && !p.hfwd && !p.self_loop;   // ... against the re-formed one: epilogue
    const int ustride = gridDim.x;
    const int u0 = blockIdx.x;
    const int niter = u0 < total ? (total - u0 + ustride - 1) / ustride : 0;
    // unit -> (graph, column block): the blocks of a graph are 8 units apart, i.e. on the same XCD at the same time
    auto unit_graph = [&](int u) { const int grp = u / (8 * nc), within = u - grp * (8 * nc); return grp * 8 + (within & 7); };
    auto unit_cb = [&](int u) { const int grp = u / (8 * nc), within = u - grp * (8 * nc); return within >> 3; };

    if (tid < 16) {          // nibble e -> bf16 (bit 0, bit 1, bit 2, bit 3) as two words
        const unsigned one = 0x3F80u;
        u32x2 v;
        v.x = ((tid & 1) ? one : 0u) | ((tid & 2) ? one << 16 : 0u);
        v.y = ((tid & 4) ? one : 0u) | ((tid & 8) ? one << 16 : 0u);
        *reinterpret_cast<u32x2*>(lut + 8 * tid) = v;
    }

    // ---- this thread's item of a tile: rows 4 rq .. 4 rq + 3, columns 4 c4 .. 4 c4 + 3 ------------------------------
    const int c4 = tid & 7, rq = tid >> 3;
    float4 v[4];
    auto issue_item = [&](int u) {        // clamped, branch-free: rows past the graph are masked at the use
        const int b = min(unit_graph(u), p.n_graphs - 1);
        const int row0 = p.node_off[b];
        const int n = max(p.node_off[b + 1] - row0, 1);
        const int col0 = unit_cb(u) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = *reinterpret_cast<const float4*>(p.x + (size_t)(row0 + min(4 * rq + k, n - 1)) * p.ldx + col0 + 4 * c4);
    };
    // prologue + split of the item (of unit u) into the planes of buffer `base`; this wave's readout / d-eps partials of
    // the unit go to parity `par`
    auto split_item = [&](int u, char* base, int par) {
        const int b = unit_graph(u);
        const bool live = b < p.n_graphs;
        const int bq = min(b, p.n_graphs - 1);
        const int row0 = p.node_off[bq];
        const int n = live ? p.node_off[bq + 1] - row0 : 0;
        const int col0 = unit_cb(u) * 32;
        const int n16 = ((n + 15) >> 4) * 16;
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
        double dot = 0.0;
        if (4 * rq < n16) {
            float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pro) {
                psc = *reinterpret_cast<const float4*>(p.p_scale + col0 + 4 * c4);
                psh = *reinterpret_cast<const float4*>(p.p_shift + col0 + 4 * c4);
            }
            const int32_t* drp = p.deg_rowptr + p.b_deg_off[bq];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = 4 * rq + k;
                float4 w = row < n ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < n) {
                    if (dot_a) {
                        const float4 hh = *reinterpret_cast<const float4*>(p.hfwd + (size_t)(row0 + row) * p.ldh + col0 + 4 * c4);
                        dot += (double)w.x * hh.x + (double)w.y * hh.y + (double)w.z * hh.z + (double)w.w * hh.w;
                    }
                    if (pro) {
                        w.x = gnm_relu(w.x * psc.x + psh.x); w.y = gnm_relu(w.y * psc.y + psh.y);
                        w.z = gnm_relu(w.z * psc.z + psh.z); w.w = gnm_relu(w.w * psc.w + psh.w);
                        if (p.p_hout) *reinterpret_cast<float4*>(p.p_hout + (size_t)(row0 + row) * p.p_ldh + col0 + 4 * c4) = w;
                        csum.x += w.x; csum.y += w.y; csum.z += w.z; csum.w += w.w;
                    }
                    if (prescale) {       // d == 0: a row nobody gathers; keep its x / 0 out of the product (see above)
                        const float d = (float)(drp[row + 1] - drp[row] + p.self_loop);
                        const bool ok = d > 0.f;
                        w.x = ok ? w.x / d : 0.f; w.y = ok ? w.y / d : 0.f; w.z = ok ? w.z / d : 0.f; w.w = ok ? w.w / d : 0.f;
                    }
                }
                v[k] = w;
            }
            const unsigned off = (unsigned)(rq >> 1) * kPK8 + (unsigned)((4 * c4 * 8 + 4 * (rq & 1)) * 2);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned x0[4], x1[4], x2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float f = c == 0 ? v[k].x : (c == 1 ? v[k].y : (c == 2 ? v[k].z : v[k].w));
                    const unsigned a1 = __float_as_uint(f) & 0xFFFF0000u;
                    const float r1 = f - __uint_as_float(a1);
                    const unsigned a2 = __float_as_uint(r1) & 0xFFFF0000u;
                    const float r2 = r1 - __uint_as_float(a2);
                    x0[k] = a1; x1[k] = a2; x2[k] = __float_as_uint(r2);
                }
                u32x2 w0, w1, w2;
                w0.x = bf16_pair_hi(x0[0], x0[1]); w0.y = bf16_pair_hi(x0[2], x0[3]);
                w1.x = bf16_pair_hi(x1[0], x1[1]); w1.y = bf16_pair_hi(x1[2], x1[3]);
                w2.x = bf16_pair_hi(x2[0], x2[1]); w2.y = bf16_pair_hi(x2[2], x2[3]);
                char* dst = base + off + c * 16;
                *reinterpret_cast<u32x2*>(dst) = w0;
                *reinterpret_cast<u32x2*>(dst + plane_bytes) = w1;
                *reinterpret_cast<u32x2*>(dst + 2u * plane_bytes) = w2;
            }
        }
        if (pro && p.p_gf) {           // readout partials of this wave: lanes with the same column chunk
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                csum.x += __shfl_xor(csum.x, off, 64); csum.y += __shfl_xor(csum.y, off, 64);
                csum.z += __shfl_xor(csum.z, off, 64); csum.w += __shfl_xor(csum.w, off, 64);
            }
            if (lane < 8) reinterpret_cast<float4*>(wscr + (par * kPWaves + wave) * kPWaveScratch)[lane] = csum;
        }
        if (dot_a) {
            const double w = wave_sum_d(dot);
            if (lane == 0) dred[(0 * 2 + par) * 16 + wave] = w;
        }
    };
    // after the hand-over that published unit u (parity par): its readout and split-side d eps, finished by one wave
    auto finish_split_side = [&](int u, int par) {
        const int b = unit_graph(u);
        if (b >= p.n_graphs) return;
        const int n = p.node_off[b + 1] - p.node_off[b];
        if (pro && p.p_gf && wave == 2 && lane < 8) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n > 0) {
                for (int w = 0; w < kPWaves; ++w) {
                    const float4 q = reinterpret_cast<const float4*>(wscr + (par * kPWaves + w) * kPWaveScratch)[lane];
                    t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                }
                if (p.p_gf_avg) {
                    const float inv = 1.f / (float)n;
                    t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
                }
            } else if (p.p_gf_avg) {
                t.x = t.y = t.z = t.w = 0.f / 0.f;
            }
            *reinterpret_cast<float4*>(p.p_gf + (size_t)b * p.p_ldgf + unit_cb(u) * 32 + 4 * lane) = t;
        }
        if (dot_a && wave == 1 && lane == 0) {
            double sdot = 0.0;
            for (int k = 0; k < kPWaves; ++k) sdot += dred[(0 * 2 + par) * 16 + k];
            p.deps_partial[(size_t)b * nc + unit_cb(u)] = n > 0 ? sdot : 0.0;
        }
    };
    // adjacency bits of this wave's row block of unit u
    auto load_bits = [&](int u, u32x4& q0, u32x4& q1) {
        const int b = min(unit_graph(u), p.n_graphs - 1);
        const int n = p.node_off[b + 1] - p.node_off[b];
        const int W = max((n + 31) >> 5, 1);
        const int HPW = aggm_half_words(W);
        const int rb = (wave + unit_cb(u)) % kPWaves;
        const uint32_t* gbits = p.adj_bits + p.b_bits_off[b];
        const u32x4* ra = reinterpret_cast<const u32x4*>(gbits + (size_t)(min(rb, W - 1) * 32 + i) * (2 * HPW) + h * HPW);
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        q0 = ra[0];
        q1 = HPW > 4 ? ra[1] : z4;
    };

    u32x4 q0 = {0u, 0u, 0u, 0u}, q1 = q0;
    if (niter > 0) {
        issue_item(u0);
        load_bits(u0, q0, q1);
        split_item(u0, smem, 0);
    }
    __syncthreads();                                       // hand-over 0: buffer 0 holds unit 0
    for (int it = 0; it < niter; ++it) {
        const int u = u0 + it * ustride;
        const int par = it & 1;
        const bool has_next = it + 1 < niter;
        GNM_PSTAMP(it, 0)
        finish_split_side(u, par);
        const char* base = smem + par * buf_bytes;
        const int b = unit_graph(u);
        const int cb = unit_cb(u);
        const bool live = b < p.n_graphs;
        const int bq = min(b, p.n_graphs - 1);
        const int row0 = p.node_off[bq];
        const int n = live ? p.node_off[bq + 1] - row0 : 0;
        const int col0 = cb * 32;
        const int col = col0 + i;
        const int W = (n + 31) >> 5;
        const int ksteps = (n + 15) >> 4;
        const int n16 = ksteps * 16;
        const int rb = (wave + cb) % kPWaves;              // rotate with the column block: evens out the SIMDs
        const bool has_rows = p.y && rb < W;
        unsigned pk[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { pk[j] = q0[j]; pk[4 + j] = q1[j]; }
        if (has_next) issue_item(u + ustride);             // the next unit's item travels under this unit's product
        const float selfB = p.self_loop ? 0.f : (p.eps ? 1.f + *p.eps : 1.f);
        const bool need_deg = AVG && !p.backward;
        float lsc = 0.f, lsh = 0.f, lmu = 0.f, s_pb = 0.f, s_ub = 0.f;
        float ss1 = 0.f, ss2 = 0.f;
        double dot = 0.0;
        if constexpr (STATS) {
            lsc = p.s_scale[col]; lsh = p.s_shift[col]; lmu = p.s_mean[col];
            if (p.s_dpool && live) {
                s_pb = p.s_dpool[(size_t)b * p.ld_dpool + col];
                if (p.s_avg) s_pb *= 1.0f / (float)max(n, 1);
            }
            if (p.s_dsc1 && live) s_ub = p.s_U[(size_t)b * p.ld_U + col];
        }
        if (has_rows) {
            const char* bp0 = base + h * kPK8 + i * 16;
            const char* bp1 = bp0 + plane_bytes;
            const char* bp2 = bp1 + plane_bytes;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            auto afrag = [&](unsigned pkw, int m) -> bf16x8 {
                const unsigned byte3 = m == 0 ? (pkw << 3) : (pkw >> (8 * m - 3));
                const unsigned lo = byte3 & 0x78u, hi = (byte3 >> 4) & 0x78u;
                const u32x2 l2 = *reinterpret_cast<const u32x2*>(lut + lo);
                const u32x2 h2 = *reinterpret_cast<const u32x2*>(lut + hi);
                const u32x4 q = {l2.x, l2.y, h2.x, h2.y};
                return __builtin_bit_cast(bf16x8, q);
            };
            auto bfrag = [&](const char* bp, int ks) -> bf16x8 {
                return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + ks * kPStep));
            };
            const int32_t* frp = p.rowptr + p.b_rp_off[b];
            const bool need_xs = prescale && !p.self_loop;
            const bool shuffled = STATS && p.s_dsc1 && row0 < p.n_batch;
            const bool has_dsc = STATS && p.s_dsc1 != nullptr;
            const float* dscp = has_dsc ? p.s_dsc1 : p.x;
            const unsigned ybytes = (unsigned)(((size_t)(n - 1) * p.ldy + p.F) * 4);
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)row0 * p.ldy, 0, (int)ybytes, 0x00020000);
            struct Ops { float zr[4], dv[4], xs[4]; int d0[4], d1[4]; };
            auto vrow_of = [&](int k, int q) { return rb * 32 + 8 * k + 4 * h + q; };
            auto request = [&](int k, Ops& o) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int vc = min(vrow_of(k, q), n - 1);
                    if constexpr (STATS) {
                        o.zr[q] = p.sZ[(size_t)(row0 + vc) * p.ldsz + col];
                        o.dv[q] = dscp[row0 + vc];
                    }
                    if constexpr (AVG) {
                        o.xs[q] = p.x[(size_t)(row0 + vc) * p.ldx + col];
                        o.d0[q] = frp[vc]; o.d1[q] = frp[vc + 1];
                    }
                }
            };
            auto finish = [&](int k, const Ops& o) {
                float ex[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ex[q] = 0.f;
                if constexpr (STATS) {
                    if (shuffled) {
                        int gq[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) gq[q] = min(max(p.s_inv_perm[min(row0 + min(vrow_of(k, q), n - 1), p.n_batch - 1)], 0), p.n_batch - 1);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float t = p.s_s2sum[gq[q]] * p.s_U[(size_t)gq[q] * p.ld_U + col];
                            ex[q] = row0 + min(vrow_of(k, q), n - 1) < p.n_batch ? t : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int vrow = vrow_of(k, q);
                    float tot = acc[4 * k + q];
                    const char* e = base + (unsigned)(min(vrow, n16 - 1) >> 3) * kPK8 + i * 16 + (vrow & 7) * 2;
                    const float e1 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e) << 16);
                    const float e2 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e + plane_bytes) << 16);
                    const float e3 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e + 2u * plane_bytes) << 16);
                    const float wv = (e1 + e2) + e3;
                    if (p.self_loop) tot += wv;
                    if constexpr (AVG) {
                        if (need_deg) tot /= (float)(o.d1[q] - o.d0[q] + p.self_loop);
                    }
                    float sb = wv;
                    if constexpr (AVG) sb = need_xs ? o.xs[q] : wv;
                    if (!p.self_loop) tot += selfB * sb;
                    if constexpr (STATS) {
                        const float zrow = o.zr[q];
                        if (want_dot && vrow < n) dot += (double)(sb * gnm_relu(zrow * lsc + lsh));
                        tot += s_pb + (has_dsc ? o.dv[q] : 0.f) * s_ub;
                        tot += ex[q];
                        if (!(zrow * lsc + lsh > 0.f)) tot = 0.f;
                        if (vrow < n) {
                            ss1 += tot;
                            ss2 += tot * (zrow - lmu);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tot), ry, (unsigned)((vrow * p.ldy + col) * 4), 0, 0);
                }
            };
            Ops oa, ob;
            request(0, oa);            // the first quarter's operands travel under the product
            GNM_PSTAMP(it, 1)
            {
                // rolling operands: a plane's fragment for step ks + 1 is requested into the register the MFMA of step
                // ks has just read (three MFMAs = 96 cycles ahead of its use: more than an LDS round trip), so the
                // product holds 12 + 8 operand registers instead of 24 + 8
                bf16x8 b0 = bfrag(bp0, 0), b1 = bfrag(bp1, 0), b2 = bfrag(bp2, 0);
                bf16x8 aA = afrag(pk[0], 0);
#pragma unroll
                for (int ks = 0; ks < 25; ++ks) {
                    if (ks < ksteps) {                                // wave-uniform
                        constexpr int LASTK = 24;
                        const int kn = ks < LASTK ? ks + 1 : LASTK;
                        const bf16x8 nA = afrag(pk[kn >> 2], kn & 3);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b0, acc, 0, 0, 0);
                        b0 = bfrag(bp0, kn);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b1, acc, 0, 0, 0);
                        b1 = bfrag(bp1, kn);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b2, acc, 0, 0, 0);
                        b2 = bfrag(bp2, kn);
                        __builtin_amdgcn_sched_barrier(0);
                        aA = nA;
                    }
                }
            }
            GNM_PSTAMP(it, 2)
            request(1, ob);
            finish(0, oa);
            request(2, oa);
            finish(1, ob);
            request(3, ob);
            finish(2, oa);
            finish(3, ob);
        }
        GNM_PSTAMP(it, 3)
        // the next unit's bit rows, into the registers the product has finished with: behind every operand request of
        // the epilogue (vector memory returns in order), in front of the split, the hand-over and the next unit's scalars
        if (has_next) load_bits(u + ustride, q0, q1);
        // ---- this wave's part of the unit's reductions, into parity scratch; finished after the hand-over
        if constexpr (STATS) {
            ss1 += __shfl_xor(ss1, 32, 64);
            ss2 += __shfl_xor(ss2, 32, 64);
            if (h == 0) {
                float* ws = reinterpret_cast<float*>(wscr + (par * kPWaves + wave) * kPWaveScratch);
                ws[i] = ss1;
                ws[32 + i] = ss2;
            }
            if (want_dot) {
                const double w = wave_sum_d(dot);
                if (lane == 0) dred[(1 * 2 + par) * 16 + wave] = w;
            }
        }
        // ---- the next unit's item has arrived under the product: prologue, split, planes of the other buffer
        GNM_PSTAMP(it, 4)
        if (has_next) split_item(u + ustride, smem + (par ^ 1) * buf_bytes, par ^ 1);
        GNM_PSTAMP(it, 5)
        __syncthreads();                                   // hand-over it + 1
        GNM_PSTAMP(it, 6)
        if constexpr (STATS) {
            if (wave == 0 && live) {
                const int which = lane >> 5, c = lane & 31;
                double sum = 0.0;
                if (n > 0) {
#pragma unroll
                    for (int w = 0; w < kPWaves; ++w)
                        sum += (double)reinterpret_cast<const float*>(wscr + (par * kPWaves + w) * kPWaveScratch)[which * 32 + c];
                    if (which) sum *= (double)p.s_rstd[col0 + c];
                }
                p.s_partial[((size_t)b * 2 + which) * 64 + col0 + c] = sum;
            }
            if (want_dot && wave == 1 && lane == 0 && live) {
                double sdot = 0.0;
                for (int k = 0; k < kPWaves; ++k) sdot += dred[(1 * 2 + par) * 16 + k];
                p.deps_partial[(size_t)b * nc + cb] = n > 0 ? sdot : 0.0;
            }
        }
    }
}

// ---- persistent form, two workgroups per CU (round 3, second design) -----------------------------------------------
// What the one-workgroup form above lacks is a second workgroup whose loads, stores and hand-overs run beside the
// first one's MFMAs; what the per-unit kernel lacks is a tile that is already there when a unit starts.  Here TWO
// 8-wave workgroups per CU each keep ONE plane buffer (as the per-unit kernel does) and stay resident: a workgroup
// requests the NEXT unit's tile into registers (two 4 x 4 items per thread, 32 registers) before it multiplies the
// current one, and splits it into the planes after its epilogue, between two barriers (everybody done reading the
// planes / the new planes visible).  To hold the tile across the product the two row blocks of a wave are multiplied
// one after the other (16 accumulator registers instead of 32; the first block's stores drain under the second
// block's product) with rolling operand fragments.
static constexpr int kQWaves = 6;                     // 12 waves per CU = 3 per SIMD: 168 registers per lane (at 4 per SIMD
                                                      // the statistics form spilled 84 with the tile in registers)
static constexpr int kQBlocks = 3;                    // row blocks per wave: role, role + 6, role + 12 (13 at n = 400)
static constexpr int kQRounds = 3;                    // 800 items / 384 threads
static constexpr int kQThreads = 64 * kQWaves;
static constexpr unsigned kQWaveScratch = 256;
static constexpr unsigned kQScratch = 128 + kQWaves * kQWaveScratch + 2 * 16 * 8;

template <bool STATS, bool AVG>
__global__ void __launch_bounds__(kQThreads, 3) gnm_aggq_kernel(const AggArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int nc = p.F >> 5;
    const int total = ((p.n_graphs + 7) >> 3) * 8 * nc;          // units, padded to whole XCD rounds
    const unsigned plane_bytes = (unsigned)(p.n16_max >> 3) * kPK8;
    char* lut = smem + 3u * plane_bytes;
    char* wscr = lut + 128;                                                    // [kQWaves][256 B]
    double* dred = reinterpret_cast<double*>(wscr + kQWaves * kQWaveScratch);  // [2 sides][16]
    const bool prescale = AVG && p.backward;
    const bool pro = !STATS && p.p_scale != nullptr;
    const bool dot_a = p.deps_partial && p.hfwd;
    const bool want_dot = STATS && p.deps_partial && !p.hfwd && !p.self_loop;
    const int ustride = gridDim.x;
    const int u0 = blockIdx.x;
    const int niter = u0 < total ? (total - u0 + ustride - 1) / ustride : 0;
    auto unit_graph = [&](int u) { const int grp = u / (8 * nc), within = u - grp * (8 * nc); return grp * 8 + (within & 7); };
    auto unit_cb = [&](int u) { const int grp = u / (8 * nc), within = u - grp * (8 * nc); return within >> 3; };

    if (tid < 16) {          // nibble e -> bf16 (bit 0, bit 1, bit 2, bit 3) as two words
        const unsigned one = 0x3F80u;
        u32x2 v;
        v.x = ((tid & 1) ? one : 0u) | ((tid & 2) ? one << 16 : 0u);
        v.y = ((tid & 4) ? one : 0u) | ((tid & 8) ? one << 16 : 0u);
        *reinterpret_cast<u32x2*>(lut + 8 * tid) = v;
    }

    // ---- this thread's items of a tile: rows 4 rq .. 4 rq + 3 (rq = tid / 8 + 48 j), columns 4 c4 .. 4 c4 + 3 ----------
    const int c4 = tid & 7, rq0 = tid >> 3;
    constexpr int RQS = kQThreads / 8;
    float4 v[kQRounds][4];
    auto issue_items = [&](int u) {       // clamped, branch-free: rows past the graph are masked at the use
        const int b = min(unit_graph(u), p.n_graphs - 1);
        const int row0 = p.node_off[b];
        const int n = max(p.node_off[b + 1] - row0, 1);
        const int col0 = unit_cb(u) * 32;
#pragma unroll
        for (int j = 0; j < kQRounds; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[j][k] = *reinterpret_cast<const float4*>(p.x + (size_t)(row0 + min(4 * (rq0 + RQS * j) + k, n - 1)) * p.ldx + col0 + 4 * c4);
    };
    auto split_items = [&](int u) {
        const int b = unit_graph(u);
        const bool live = b < p.n_graphs;
        const int bq = min(b, p.n_graphs - 1);
        const int row0 = p.node_off[bq];
        const int n = live ? p.node_off[bq + 1] - row0 : 0;
        const int col0 = unit_cb(u) * 32;
        const int n16 = ((n + 15) >> 4) * 16;
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
        double dot = 0.0;
        float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pro) {
            psc = *reinterpret_cast<const float4*>(p.p_scale + col0 + 4 * c4);
            psh = *reinterpret_cast<const float4*>(p.p_shift + col0 + 4 * c4);
        }
        const int32_t* drp = p.deg_rowptr + p.b_deg_off[bq];
#pragma unroll
        for (int j = 0; j < kQRounds; ++j) {
            const int rq = rq0 + RQS * j;
            if (4 * rq < n16) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = 4 * rq + k;
                    float4 w = row < n ? v[j][k] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < n) {
                        if (dot_a) {
                            const float4 hh = *reinterpret_cast<const float4*>(p.hfwd + (size_t)(row0 + row) * p.ldh + col0 + 4 * c4);
                            dot += (double)w.x * hh.x + (double)w.y * hh.y + (double)w.z * hh.z + (double)w.w * hh.w;
                        }
                        if (pro) {
                            w.x = gnm_relu(w.x * psc.x + psh.x); w.y = gnm_relu(w.y * psc.y + psh.y);
                            w.z = gnm_relu(w.z * psc.z + psh.z); w.w = gnm_relu(w.w * psc.w + psh.w);
                            if (p.p_hout) *reinterpret_cast<float4*>(p.p_hout + (size_t)(row0 + row) * p.p_ldh + col0 + 4 * c4) = w;
                            csum.x += w.x; csum.y += w.y; csum.z += w.z; csum.w += w.w;
                        }
                        if (prescale) {
                            const float d = (float)(drp[row + 1] - drp[row] + p.self_loop);
                            const bool ok = d > 0.f;
                            w.x = ok ? w.x / d : 0.f; w.y = ok ? w.y / d : 0.f; w.z = ok ? w.z / d : 0.f; w.w = ok ? w.w / d : 0.f;
                        }
                    }
                    v[j][k] = w;
                }
                const unsigned off = (unsigned)(rq >> 1) * kPK8 + (unsigned)((4 * c4 * 8 + 4 * (rq & 1)) * 2);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned x0[4], x1[4], x2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float f = c == 0 ? v[j][k].x : (c == 1 ? v[j][k].y : (c == 2 ? v[j][k].z : v[j][k].w));
                        const unsigned a1 = __float_as_uint(f) & 0xFFFF0000u;
                        const float r1 = f - __uint_as_float(a1);
                        const unsigned a2 = __float_as_uint(r1) & 0xFFFF0000u;
                        const float r2 = r1 - __uint_as_float(a2);
                        x0[k] = a1; x1[k] = a2; x2[k] = __float_as_uint(r2);
                    }
                    u32x2 w0, w1, w2;
                    w0.x = bf16_pair_hi(x0[0], x0[1]); w0.y = bf16_pair_hi(x0[2], x0[3]);
                    w1.x = bf16_pair_hi(x1[0], x1[1]); w1.y = bf16_pair_hi(x1[2], x1[3]);
                    w2.x = bf16_pair_hi(x2[0], x2[1]); w2.y = bf16_pair_hi(x2[2], x2[3]);
                    char* dst = smem + off + c * 16;
                    *reinterpret_cast<u32x2*>(dst) = w0;
                    *reinterpret_cast<u32x2*>(dst + plane_bytes) = w1;
                    *reinterpret_cast<u32x2*>(dst + 2u * plane_bytes) = w2;
                }
            }
        }
        if (pro && p.p_gf) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                csum.x += __shfl_xor(csum.x, off, 64); csum.y += __shfl_xor(csum.y, off, 64);
                csum.z += __shfl_xor(csum.z, off, 64); csum.w += __shfl_xor(csum.w, off, 64);
            }
            if (lane < 8) reinterpret_cast<float4*>(wscr + wave * kQWaveScratch)[lane] = csum;
        }
        if (dot_a) {
            const double w = wave_sum_d(dot);
            if (lane == 0) dred[wave] = w;
        }
    };
    // behind the barrier that published unit u's planes: its readout and split-side d eps, finished by one wave
    auto finish_split_side = [&](int u) {
        const int b = unit_graph(u);
        if (b >= p.n_graphs) return;
        const int n = p.node_off[b + 1] - p.node_off[b];
        if (pro && p.p_gf && wave == 2 && lane < 8) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n > 0) {
                for (int w = 0; w < kQWaves; ++w) {
                    const float4 q = reinterpret_cast<const float4*>(wscr + w * kQWaveScratch)[lane];
                    t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                }
                if (p.p_gf_avg) {
                    const float inv = 1.f / (float)n;
                    t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
                }
            } else if (p.p_gf_avg) {
                t.x = t.y = t.z = t.w = 0.f / 0.f;
            }
            *reinterpret_cast<float4*>(p.p_gf + (size_t)b * p.p_ldgf + unit_cb(u) * 32 + 4 * lane) = t;
        }
        if (dot_a && wave == 1 && lane == 0) {
            double sdot = 0.0;
            for (int k = 0; k < kQWaves; ++k) sdot += dred[k];
            p.deps_partial[(size_t)b * nc + unit_cb(u)] = n > 0 ? sdot : 0.0;
        }
    };
    // adjacency bits of this wave's row blocks of unit u
    auto load_bits = [&](int u, u32x4 (&q)[2 * kQBlocks]) {
        const int b = min(unit_graph(u), p.n_graphs - 1);
        const int n = p.node_off[b + 1] - p.node_off[b];
        const int W = max((n + 31) >> 5, 1);
        const int HPW = aggm_half_words(W);
        const int role = (wave + unit_cb(u)) % kQWaves;
        const uint32_t* gbits = p.adj_bits + p.b_bits_off[b];
        const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < kQBlocks; ++t) {
            const u32x4* ra = reinterpret_cast<const u32x4*>(gbits + (size_t)(min(role + kQWaves * t, W - 1) * 32 + i) * (2 * HPW) + h * HPW);
            q[2 * t] = ra[0];
            q[2 * t + 1] = HPW > 4 ? ra[1] : z4;
        }
    };

    u32x4 q[2 * kQBlocks];
    {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 2 * kQBlocks; ++t) q[t] = z4;
    }
    if (niter > 0) {
        issue_items(u0);
        load_bits(u0, q);
        split_items(u0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int u = u0 + it * ustride;
        const bool has_next = it + 1 < niter;
        finish_split_side(u);
        const int b = unit_graph(u);
        const int cb = unit_cb(u);
        const bool live = b < p.n_graphs;
        const int bq = min(b, p.n_graphs - 1);
        const int row0 = p.node_off[bq];
        const int n = live ? p.node_off[bq + 1] - row0 : 0;
        const int col0 = cb * 32;
        const int col = col0 + i;
        const int W = (n + 31) >> 5;
        const int ksteps = (n + 15) >> 4;
        const int n16 = ksteps * 16;
        const int role = (wave + cb) % kQWaves;
        if (has_next) issue_items(u + ustride);            // the next unit's tile travels under this unit's products
        const float selfB = p.self_loop ? 0.f : (p.eps ? 1.f + *p.eps : 1.f);
        const bool need_deg = AVG && !p.backward;
        float lsc = 0.f, lsh = 0.f, lmu = 0.f, s_pb = 0.f, s_ub = 0.f;
        float ss1 = 0.f, ss2 = 0.f;
        double dot = 0.0;
        if constexpr (STATS) {
            lsc = p.s_scale[col]; lsh = p.s_shift[col]; lmu = p.s_mean[col];
            if (p.s_dpool && live) {
                s_pb = p.s_dpool[(size_t)b * p.ld_dpool + col];
                if (p.s_avg) s_pb *= 1.0f / (float)max(n, 1);
            }
            if (p.s_dsc1 && live) s_ub = p.s_U[(size_t)b * p.ld_U + col];
        }
        const char* bp0 = smem + h * kPK8 + i * 16;
        const char* bp1 = bp0 + plane_bytes;
        const char* bp2 = bp1 + plane_bytes;
        const int32_t* frp = p.rowptr + p.b_rp_off[bq];
        const bool need_xs = prescale && !p.self_loop;
        const bool shuffled = STATS && p.s_dsc1 && row0 < p.n_batch;
        const bool has_dsc = STATS && p.s_dsc1 != nullptr;
        const float* dscp = has_dsc ? p.s_dsc1 : p.x;
        const unsigned ybytes = (unsigned)(((size_t)max(n - 1, 0) * p.ldy + p.F) * 4);
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)row0 * p.ldy, 0, (int)ybytes, 0x00020000);
        auto afrag = [&](unsigned pkw, int m) -> bf16x8 {
            const unsigned byte3 = m == 0 ? (pkw << 3) : (pkw >> (8 * m - 3));
            const unsigned lo = byte3 & 0x78u, hi = (byte3 >> 4) & 0x78u;
            const u32x2 l2 = *reinterpret_cast<const u32x2*>(lut + lo);
            const u32x2 h2 = *reinterpret_cast<const u32x2*>(lut + hi);
            const u32x4 qq = {l2.x, l2.y, h2.x, h2.y};
            return __builtin_bit_cast(bf16x8, qq);
        };
        auto bfrag = [&](const char* bp, int ks) -> bf16x8 {
            return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + ks * kPStep));
        };
        struct Ops { float zr[4], dv[4], xs[4]; int d0[4], d1[4]; };
        // one 32-row block: product on the rolling fragments, then the epilogue from the accumulator
        auto block = [&](int rb, const u32x4& qa, const u32x4& qb) {
            unsigned pk[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { pk[j] = qa[j]; pk[4 + j] = qb[j]; }
            auto vrow_of = [&](int k, int qq) { return rb * 32 + 8 * k + 4 * h + qq; };
            auto request = [&](int k, Ops& o) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int vc = min(vrow_of(k, qq), n - 1);
                    if constexpr (STATS) {
                        o.zr[qq] = p.sZ[(size_t)(row0 + vc) * p.ldsz + col];
                        o.dv[qq] = dscp[row0 + vc];
                    }
                    if constexpr (AVG) {
                        o.xs[qq] = p.x[(size_t)(row0 + vc) * p.ldx + col];
                        o.d0[qq] = frp[vc]; o.d1[qq] = frp[vc + 1];
                    }
                }
            };
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            auto finish = [&](int k, const Ops& o) {
                float ex[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) ex[qq] = 0.f;
                if constexpr (STATS) {
                    if (shuffled) {
                        int gq[4];
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) gq[qq] = min(max(p.s_inv_perm[min(row0 + min(vrow_of(k, qq), n - 1), p.n_batch - 1)], 0), p.n_batch - 1);
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const float t = p.s_s2sum[gq[qq]] * p.s_U[(size_t)gq[qq] * p.ld_U + col];
                            ex[qq] = row0 + min(vrow_of(k, qq), n - 1) < p.n_batch ? t : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int vrow = vrow_of(k, qq);
                    float tot = acc[4 * k + qq];
                    const char* e = smem + (unsigned)(min(vrow, n16 - 1) >> 3) * kPK8 + i * 16 + (vrow & 7) * 2;
                    const float e1 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e) << 16);
                    const float e2 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e + plane_bytes) << 16);
                    const float e3 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e + 2u * plane_bytes) << 16);
                    const float wv = (e1 + e2) + e3;
                    if (p.self_loop) tot += wv;
                    if constexpr (AVG) {
                        if (need_deg) tot /= (float)(o.d1[qq] - o.d0[qq] + p.self_loop);
                    }
                    float sb = wv;
                    if constexpr (AVG) sb = need_xs ? o.xs[qq] : wv;
                    if (!p.self_loop) tot += selfB * sb;
                    if constexpr (STATS) {
                        const float zrow = o.zr[qq];
                        if (want_dot && vrow < n) dot += (double)(sb * gnm_relu(zrow * lsc + lsh));
                        tot += s_pb + (has_dsc ? o.dv[qq] : 0.f) * s_ub;
                        tot += ex[qq];
                        if (!(zrow * lsc + lsh > 0.f)) tot = 0.f;
                        if (vrow < n) {
                            ss1 += tot;
                            ss2 += tot * (zrow - lmu);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tot), ry, (unsigned)((vrow * p.ldy + col) * 4), 0, 0);
                }
            };
            Ops oa, ob;
            request(0, oa);
            {
                bf16x8 b0 = bfrag(bp0, 0), b1 = bfrag(bp1, 0), b2 = bfrag(bp2, 0);
                bf16x8 aA = afrag(pk[0], 0);
#pragma unroll
                for (int ks = 0; ks < 25; ++ks) {
                    if (ks < ksteps) {                                // wave-uniform
                        constexpr int LASTK = 24;
                        const int kn = ks < LASTK ? ks + 1 : LASTK;
                        const bf16x8 nA = afrag(pk[kn >> 2], kn & 3);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b0, acc, 0, 0, 0);
                        b0 = bfrag(bp0, kn);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b1, acc, 0, 0, 0);
                        b1 = bfrag(bp1, kn);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b2, acc, 0, 0, 0);
                        b2 = bfrag(bp2, kn);
                        __builtin_amdgcn_sched_barrier(0);
                        aA = nA;
                    }
                }
            }
            request(1, ob);
            finish(0, oa);
            request(2, oa);
            finish(1, ob);
            request(3, ob);
            finish(2, oa);
            finish(3, ob);
        };
#pragma unroll
        for (int t = 0; t < kQBlocks; ++t)
            if (p.y && role + kQWaves * t < W) block(role + kQWaves * t, q[2 * t], q[2 * t + 1]);
        if (has_next) load_bits(u + ustride, q);           // behind every operand request of the epilogues
        if constexpr (STATS) {
            ss1 += __shfl_xor(ss1, 32, 64);
            ss2 += __shfl_xor(ss2, 32, 64);
            if (h == 0) {
                float* ws = reinterpret_cast<float*>(wscr + wave * kQWaveScratch);
                ws[i] = ss1;
                ws[32 + i] = ss2;
            }
            if (want_dot) {
                const double w = wave_sum_d(dot);
                if (lane == 0) dred[16 + wave] = w;
            }
        }
        __syncthreads();                 // everybody is done reading the planes; the unit's partial sums are in scratch
        if constexpr (STATS) {
            if (wave == 0 && live) {
                const int which = lane >> 5, c = lane & 31;
                double sum = 0.0;
                if (n > 0) {
#pragma unroll
                    for (int w = 0; w < kQWaves; ++w)
                        sum += (double)reinterpret_cast<const float*>(wscr + w * kQWaveScratch)[which * 32 + c];
                    if (which) sum *= (double)p.s_rstd[col0 + c];
                }
                p.s_partial[((size_t)b * 2 + which) * 64 + col0 + c] = sum;
            }
            if (want_dot && wave == 1 && lane == 0 && live) {
                double sdot = 0.0;
                for (int k = 0; k < kQWaves; ++k) sdot += dred[16 + k];
                p.deps_partial[(size_t)b * nc + cb] = n > 0 ? sdot : 0.0;
            }
        }
        if (has_next) split_items(u + ustride);
        __syncthreads();                 // the next unit's planes (and its readout partials) are visible
    }
}

// ---- bit adjacency ----------------------------------------------------------------------------
// graph g: W = ceil(n / 32) bit words per row = 4 W bytes; byte j of a row holds columns 8 j .. 8 j + 7 (bit k % 8 of
// byte k / 8 = 1 iff k is in row v of the CSR).  The bytes are stored DE-INTERLEAVED: even bytes (j = 2 s: the first 8
// columns of 16-column MFMA step s) in the row's first half, odd bytes in its second, each half padded with zeros to
// HP = ceil(W / 2) rounded up to 4 words; a row is 2 HP words, there are 32 W rows (zero rows pad the last 32-row
// block).  So  byte j -> half j & 1, position j >> 1.  dup[g] = number of CSR entries that hit a bit already set (a
// multigraph's repeated edge: the bit matrix cannot carry its weight -- the caller keeps such graphs on the CSR path).
extern "C" long long gnm_adj_bits_words(int n) {
    const long long W = (n + 31) / 32;
    return W * 32 * 2 * aggm_half_words((int)W);
}
extern "C" int gnm_aggm_max_nodes(void) { return kAggmMaxN; }
extern "C" int gnm_aggm_num_partials(int F, int B) { return (F % 32) ? 0 : B * (F / 32); }   // (F < 32: no d-eps form)

__global__ void __launch_bounds__(256) gnm_adj_bits_build_kernel(const int32_t* rowptr, const uint16_t* colv,
                                                                 const int64_t* g_rp_off, const int64_t* g_col_off,
                                                                 const int32_t* g_n, uint32_t* bits,
                                                                 const int64_t* g_bits_off, int32_t* dup) {
    const int g = blockIdx.x;
    const int n = g_n[g];
    const int W = (n + 31) >> 5, HP = aggm_half_words(W), RPW = 2 * HP;
    uint32_t* out = bits + g_bits_off[g];
    const int words = W * 32 * RPW;
    for (int k = threadIdx.x; k < words; k += blockDim.x) out[k] = 0u;
    __syncthreads();
    const int32_t* rp = rowptr + g_rp_off[g];
    const uint16_t* cl = colv + g_col_off[g];
    int ndup = 0;
    for (int v = threadIdx.x >> 3; v < n; v += blockDim.x >> 3) {        // 8 threads per row
        for (int e = rp[v] + (threadIdx.x & 7); e < rp[v + 1]; e += 8) {
            const unsigned k = cl[e];
            const unsigned j = k >> 3, pos = j >> 1;                     // byte of the row, its place in its half
            const unsigned word = (j & 1) * HP + (pos >> 2);
            const unsigned bit = 1u << ((pos & 3) * 8 + (k & 7));
            const unsigned old = atomicOr(out + (size_t)v * RPW + word, bit);
            ndup += (old & bit) ? 1 : 0;
        }
    }
    if (ndup) atomicAdd(dup + g, ndup);
}

extern "C" int gnm_adj_bits_build(const int32_t* rowptr, const uint16_t* col, const int64_t* g_rp_off,
                                  const int64_t* g_col_off, const int32_t* g_n, int G, uint32_t* bits,
                                  const int64_t* g_bits_off, int32_t* dup, void* stream) {
    if (G <= 0) return GNM_OK;
    if (!rowptr || !g_rp_off || !g_col_off || !g_n || !bits || !g_bits_off || !dup) return GNM_ERR_BAD_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GNM_HIP(hipMemsetAsync(dup, 0, (size_t)G * 4, s));
    hipLaunchKernelGGL(gnm_adj_bits_build_kernel, dim3(G), dim3(256), 0, s, rowptr, col, g_rp_off, g_col_off, g_n, bits,
                       g_bits_off, dup);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---- launch -------------------------------------------------------------------------------------
// Launches of at least this many units (graphs x 32-column blocks) take the persistent kernel.  v >= 0 sets the
// threshold (tests run every shape through both kernels; 1 = always), v < 0 only reads it; returns the previous value.
// Initial value: GNM_AGGP_MIN_UNITS, else OFF (1 << 30): measured on MI355X at B = 1024 the persistent form is SLOWER
// than the workgroup-per-unit kernel -- 105 / 98 / 144 us against 86 / 92 / 121 us (plain / fused prologue / backward
// with statistics; profiles/r03_aggp_timeline.md).  Its in-kernel timeline says why: with one workgroup per CU every
// wave is in the same phase, so the matrix pipe idles through epilogue, split and hand-over (product 31 % of a unit,
// 26 % of the median wave's time spent waiting at the hand-over for the SIMD that holds 4 of the 13 row blocks), where
// two independent workgroups per CU overlap one's product with the other's loads and stores.  Kept, tested in both
// forms, as the base for a two-workgroup variant.
extern "C" int gnm_aggm_persistent_min_units(int v) {
    static int value = gnm_env_int("GNM_AGGP_MIN_UNITS", 1 << 30);
    const int old = value;
    if (v >= 0) value = v;
    return old;
}

// Which persistent kernel the launches above the threshold take: 1 = one 13-wave workgroup per CU with two plane buffers
// (gnm_aggp_kernel), 2 = two 8-wave workgroups per CU with the next tile held in registers (gnm_aggq_kernel).
// v = 1 / 2 selects, anything else only reads; returns the previous value.  Initial value: GNM_AGGM_FORM or 2.
extern "C" int gnm_aggm_persistent_form(int v) {
    static int value = gnm_env_int("GNM_AGGM_FORM", 2) == 1 ? 1 : 2;
    const int old = value;
    if (v == 1 || v == 2) value = v;
    return old;
}

static bool aggm_shape_ok(const AggArgs& a, int n_max) {
    if (!a.adj_bits || !a.b_bits_off || (reinterpret_cast<uintptr_t>(a.adj_bits) & 15)) return false;
    if (n_max < 1 || n_max > kAggmMaxN) return false;
    const bool narrow = a.F < 32;                  // one partial column block: plain form only
    if (narrow && (a.sZ || a.p_scale || a.deps_partial)) return false;
    if (!narrow && ((a.F & 31) || a.F > 256)) return false;
    if (!narrow && ((a.ldx & 3) || (reinterpret_cast<uintptr_t>(a.x) & 15))) return false;
    if (a.hfwd && ((a.ldh & 3) || (reinterpret_cast<uintptr_t>(a.hfwd) & 15))) return false;
    return true;
}

static int launch_aggm(AggArgs a, int B, int n_max, bool stats, hipStream_t stream) {
#ifdef GNM_AGG16_TUNING
    a.stamps = g_aggm_stamps;
#else
    a.stamps = nullptr;
#endif
    a.n_graphs = B;
    a.n16_max = ((n_max + 15) / 16) * 16;
    const size_t lds = (size_t)3 * (a.n16_max / 8) * kAggmK8Stride + kAggmScratch;
    const int nc = a.F < 32 ? 1 : a.F / 32;
    const int grid = ((B + 7) / 8) * 8 * nc;
    // the persistent form (one workgroup per CU, loader + compute waves, two plane buffers): whole column blocks,
    // graphs of at most 400 nodes, and enough units that every CU gets a few (below that the per-unit kernel's many
    // small workgroups fill the chip better).  GNM_AGGM_NO_PERSIST=1 / GNM_AGGP_MIN_UNITS: A/B knobs, read once.
    static const int no_persist = gnm_env_int("GNM_AGGM_NO_PERSIST", 0);
    const int min_units = gnm_aggm_persistent_min_units(-1);
    // (the statistics form for neighbour "average" spills at 128 registers in the persistent kernel: per-unit kernel)
    if (!no_persist && a.F >= 32 && n_max <= kPMaxN && grid >= min_units &&
        !(stats && a.average && gnm_aggm_persistent_form(0) == 1)) {
        int dev = 0, cus = 0;
        GNM_HIP(hipGetDevice(&dev));
        GNM_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (gnm_aggm_persistent_form(0) == 2) {            // two workgroups per CU, one plane buffer each
            int qgrid = (2 * cus / (8 * nc)) * (8 * nc);
            if (qgrid <= 0) qgrid = 8 * nc;
            if (qgrid > grid) qgrid = grid;
            const size_t qlds = (size_t)3 * (a.n16_max / 8) * kPK8 + kQScratch;
            if (2 * qlds > (size_t)kLdsBudget) return GNM_ERR_BAD_ARG;
#define GNM_AGGQ_LAUNCH(S_, A_)                                                                              \
    do {                                                                                                     \
        GNM_ALLOW_FULL_LDS((&gnm_aggq_kernel<S_, A_>));                                                      \
        hipLaunchKernelGGL((gnm_aggq_kernel<S_, A_>), dim3(qgrid), dim3(kQThreads), qlds, stream, a);        \
    } while (0)
            if (stats && a.average) GNM_AGGQ_LAUNCH(true, true);
            else if (stats) GNM_AGGQ_LAUNCH(true, false);
            else if (a.average) GNM_AGGQ_LAUNCH(false, true);
            else GNM_AGGQ_LAUNCH(false, false);
#undef GNM_AGGQ_LAUNCH
            GNM_CHECK_LAUNCH();
            return GNM_OK;
        }
        int pgrid = (cus / (8 * nc)) * (8 * nc);           // keeps a graph's column blocks on one XCD, in step
        if (pgrid <= 0) pgrid = 8 * nc;
        if (pgrid > grid) pgrid = grid;
        const size_t plds = (size_t)2 * 3 * (a.n16_max / 8) * kPK8 + kPScratch;
        if (plds > (size_t)kLdsBudget) return GNM_ERR_BAD_ARG;
#define GNM_AGGP_LAUNCH(S_, A_)                                                                              \
    do {                                                                                                     \
        GNM_ALLOW_FULL_LDS((&gnm_aggp_kernel<S_, A_>));                                                      \
        hipLaunchKernelGGL((gnm_aggp_kernel<S_, A_>), dim3(pgrid), dim3(kPThreads), plds, stream, a);        \
    } while (0)
        if (stats && a.average) GNM_AGGP_LAUNCH(true, true);
        else if (stats) GNM_AGGP_LAUNCH(true, false);
        else if (a.average) GNM_AGGP_LAUNCH(false, true);
        else GNM_AGGP_LAUNCH(false, false);
#undef GNM_AGGP_LAUNCH
        GNM_CHECK_LAUNCH();
        return GNM_OK;
    }
#define GNM_AGGM_LAUNCH(S_, A_)                                                                              \
    do {                                                                                                     \
        GNM_ALLOW_FULL_LDS((&gnm_aggm_kernel<S_, A_>));                                                      \
        hipLaunchKernelGGL((gnm_aggm_kernel<S_, A_>), dim3(grid), dim3(kAggmThreads), lds, stream, a);       \
    } while (0)
    if (a.F < 32) {
        GNM_ALLOW_FULL_LDS((&gnm_aggm_kernel<false, true, true>));
        GNM_ALLOW_FULL_LDS((&gnm_aggm_kernel<false, false, true>));
        if (a.average) hipLaunchKernelGGL((gnm_aggm_kernel<false, true, true>), dim3(grid), dim3(kAggmThreads), lds, stream, a);
        else hipLaunchKernelGGL((gnm_aggm_kernel<false, false, true>), dim3(grid), dim3(kAggmThreads), lds, stream, a);
    } else if (stats && a.average) GNM_AGGM_LAUNCH(true, true);
    else if (stats) GNM_AGGM_LAUNCH(true, false);
    else if (a.average) GNM_AGGM_LAUNCH(false, true);
    else GNM_AGGM_LAUNCH(false, false);
#undef GNM_AGGM_LAUNCH
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// gnm_agg over the bit adjacency.  Same arguments as gnm_agg plus (adj_bits, b_bits_off); the CSR arguments keep
// their meaning (degrees come from them).  GNM_ERR_UNSUPPORTED: shape outside this kernel (n_max > 416, F not a
// multiple of 32, unaligned rows) -- call gnm_agg.  deps_partial receives gnm_aggm_num_partials(F, B) values.
extern "C" int gnm_aggm(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off, const int64_t* b_col_off,
                        const uint32_t* adj_bits, const int64_t* b_bits_off, const int32_t* deg_rowptr,
                        const int64_t* b_deg_off, const int32_t* node_off, int B, int n_max, const float* x, int ldx,
                        float* y, int ldy, int F, const float* eps, int average, int self_loop, int backward,
                        const float* hfwd, int ldh, double* deps_partial, void* stream) {
    if (B <= 0) return GNM_OK;
    if (F <= 0 || n_max < 0) return GNM_ERR_BAD_ARG;
    if (deps_partial && !hfwd) return GNM_ERR_BAD_ARG;
    AggArgs a;
    memset(&a, 0, sizeof(a));
    a.rowptr = rowptr; a.col = col; a.b_rp_off = b_rp_off; a.b_col_off = b_col_off;
    a.adj_bits = adj_bits; a.b_bits_off = b_bits_off;
    a.deg_rowptr = deg_rowptr ? deg_rowptr : rowptr;
    a.b_deg_off = b_deg_off ? b_deg_off : b_rp_off;
    a.node_off = node_off; a.x = x; a.y = y; a.eps = eps; a.hfwd = hfwd; a.deps_partial = deps_partial;
    a.ldx = ldx; a.ldy = ldy; a.ldh = ldh; a.F = F; a.nslices = 1;
    a.average = average; a.self_loop = self_loop; a.backward = backward;
    if (!aggm_shape_ok(a, n_max)) return GNM_ERR_UNSUPPORTED;
    return launch_aggm(a, B, n_max, false, reinterpret_cast<hipStream_t>(stream));
}

// gnm_agg_bwd_stats over the bit adjacency (F == 64 only, like the CSR form).
extern "C" int gnm_aggm_bwd_stats(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off,
                                  const int64_t* b_col_off, const uint32_t* adj_bits, const int64_t* b_bits_off,
                                  const int32_t* deg_rowptr, const int64_t* b_deg_off, const int32_t* node_off, int B,
                                  int n_max, const float* x, int ldx, float* y, int ldy, int F, const float* eps,
                                  int average, int self_loop, const float* hfwd, int ldh, double* deps_partial,
                                  const float* sZ, int ldsz, const float* s_scale, const float* s_shift,
                                  const float* s_mean, const float* s_rstd, const float* dpool, int ld_dpool,
                                  int graph_avg, const float* dsc1, const float* U, int ld_U, const int32_t* inv_perm,
                                  const float* s2sum, double* s_partial, void* stream) {
    if (B <= 0) return GNM_OK;
    if (F != 64 || !y || !sZ || !s_partial) return GNM_ERR_UNSUPPORTED;
    if (deps_partial && !hfwd && self_loop) return GNM_ERR_BAD_ARG;
    AggArgs a;
    memset(&a, 0, sizeof(a));
    a.rowptr = rowptr; a.col = col; a.b_rp_off = b_rp_off; a.b_col_off = b_col_off;
    a.adj_bits = adj_bits; a.b_bits_off = b_bits_off;
    a.deg_rowptr = deg_rowptr ? deg_rowptr : rowptr;
    a.b_deg_off = b_deg_off ? b_deg_off : b_rp_off;
    a.node_off = node_off; a.x = x; a.y = y; a.eps = eps; a.hfwd = hfwd; a.deps_partial = deps_partial;
    a.ldx = ldx; a.ldy = ldy; a.ldh = ldh; a.F = F; a.nslices = 1;
    a.average = average; a.self_loop = self_loop; a.backward = 1;
    a.sZ = sZ; a.s_scale = s_scale; a.s_shift = s_shift; a.s_mean = s_mean; a.s_rstd = s_rstd;
    a.s_dpool = dpool; a.s_dsc1 = dsc1; a.s_U = U; a.s_inv_perm = inv_perm; a.s_s2sum = s2sum;
    a.s_partial = s_partial; a.ldsz = ldsz; a.ld_dpool = ld_dpool; a.ld_U = ld_U; a.s_avg = graph_avg;
    a.n_batch = B;
    if (!aggm_shape_ok(a, n_max)) return GNM_ERR_UNSUPPORTED;
    return launch_aggm(a, B, n_max, true, reinterpret_cast<hipStream_t>(stream));
}

// gnm_agg_fwd_bnrelu over the bit adjacency (F == 64 only, like the CSR form).
extern "C" int gnm_aggm_fwd_bnrelu(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off,
                                   const int64_t* b_col_off, const uint32_t* adj_bits, const int64_t* b_bits_off,
                                   const int32_t* node_off, int B, int n_max, const float* z, int ldz,
                                   const float* scale, const float* shift, float* hout, int ldh, float* gf, int ldgf,
                                   int graph_avg, float* y, int ldy, int F, const float* eps, int average,
                                   int self_loop, void* stream) {
    if (B <= 0) return GNM_OK;
    if (F != 64 || !y || !z || !scale || !shift) return GNM_ERR_UNSUPPORTED;
    if ((hout && (ldh & 3)) || (gf && (ldgf & 3))) return GNM_ERR_UNSUPPORTED;
    const uintptr_t al = reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
                         reinterpret_cast<uintptr_t>(hout) | reinterpret_cast<uintptr_t>(gf);
    if (al & 15) return GNM_ERR_UNSUPPORTED;
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
    if (!aggm_shape_ok(a, n_max)) return GNM_ERR_UNSUPPORTED;
    return launch_aggm(a, B, n_max, false, reinterpret_cast<hipStream_t>(stream));
}
