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
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

#ifndef GNM_AGGM_WAVES
#define GNM_AGGM_WAVES 8
#endif
static constexpr int kAggmWaves = GNM_AGGM_WAVES;   // row blocks w and w + kAggmWaves per wave (13 blocks at n = 400)
static constexpr int kAggmThreads = 64 * kAggmWaves;
static constexpr int kAggmMaxN = 416;            // 13 row blocks; up to 400 nodes two workgroups share a CU's LDS
// nibble table + (forward prologue forms) readout partials / (statistics form) the graph's per-row discriminator gradient
static constexpr int kAggmScratch = 128 + 4 * kAggmMaxN;
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
    GNM_MSTAMP(8)
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
    float* dvs = reinterpret_cast<float*>(lut + 128);          // STATS only (rsum is the forward prologue's)
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
    if (tid < 16) {          // nibble e -> bf16 (bit 0, bit 1, bit 2, bit 3) as two words
        const unsigned one = 0x3F80u;
        u32x2 v;
        v.x = ((tid & 1) ? one : 0u) | ((tid & 2) ? one << 16 : 0u);
        v.y = ((tid & 4) ? one : 0u) | ((tid & 8) ? one << 16 : 0u);
        *reinterpret_cast<u32x2*>(lut + 8 * tid) = v;
    }

    // Statistics form: the per-row gradient of the discriminator's first score (one float per node) goes to LDS, one
    // coalesced load, instead of sixteen 4-byte global loads per row block and lane in the epilogue.  Requested here,
    // unconditionally (absent: any readable address); written to LDS behind the tile (a write here would wait for it).
    float dv_early = 0.f;
    if constexpr (STATS) dv_early = (p.s_dsc1 ? p.s_dsc1 : p.x)[row0 + min(tid, n - 1)];

    // ---- phase A ------------------------------------------------------------------------------
    // item = (4 consecutive rows, 4 consecutive columns): four 16-B loads, transposed in registers into 8-B pieces
    // of the planes (4 consecutive k of one column)
    // Which item a lane takes (round 3): a wave covers 8 row quads x 8 column chunks per round as before -- the same
    // set of global addresses per load instruction -- but with the chunk's low bit in lane bit 0, the row quad in lane
    // bits 1-3 and the chunk's high bits in lane bits 4-5.  The 16 lanes of one ds_write_b64 group then differ in
    // (chunk & 1) -> +16 banks, (quad & 1) -> +2 banks, (quad >> 1) -> +4 banks per 528-byte k-group: 16 distinct bank
    // pairs.  With chunk = lane & 7 they fell on 4 bank pairs (4-way conflicts on every write: the
    // SQ_LDS_BANK_CONFLICT = 1.135e7 per launch, identical in all three launch forms, that VERDICT r2 asked about).
    const int c4 = (lane & 1) | ((lane >> 4) << 1);
    const int rql = (lane >> 1) & 7;
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pro) {
        psc = *reinterpret_cast<const float4*>(p.p_scale + col0 + 4 * c4);
        psh = *reinterpret_cast<const float4*>(p.p_shift + col0 + 4 * c4);
    }
    double dot = 0.0;
    constexpr int UA = (832 + kAggmThreads - 1) / kAggmThreads;   // 416 / 4 * 8 = 832 items
    float4 v[UA][4];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int rq = ((wave + u * kAggmWaves) << 3) | rql;
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
    // (round 4: requested BEHIND the tile loads.  Their address needs one more dependent scalar read -- the graph's
    //  offset into the bit arena, ~1.8 us after the node offsets under load per the in-kernel timeline -- which now
    //  passes while the tile is on its way from HBM; the bits themselves mostly come from L2)
    __builtin_amdgcn_sched_barrier(0);          // (left alone, the scheduler hoists that scalar read and its wait above the tile loads)
    // Bit rows (round 3 layout, see gnm_adj_bits_build): a row's bytes are stored de-interleaved -- the even bytes
    // (columns 16 s .. 16 s + 7 of MFMA step s: what lanes 0-31 multiply) in the first half of the row, the odd bytes
    // (columns 16 s + 8 .. + 15: lanes 32-63) in the second, each half padded to 16-byte pieces.  A lane therefore
    // loads exactly the bytes it uses, byte m of word j = step 4 j + m, as one or two 16-byte pieces: no permutes
    // and half the registers of the interleaved layout (which every lane had to load whole).
    const int HPW = (((W + 1) >> 1) + 3) & ~3;                   // words per half row (4 or 8)
    unsigned pkA[8], pkB[8];
    // (1 + eps): requested first and consumed in the epilogue.  Unconditional (an absent eps reads a valid address and
    // is dropped): a load inside a branch is drained right there, and until round 4 this one sat behind the barrier
    // with its own vmcnt(0) -- an L2 round trip on every workgroup's critical path.
    const float eps_raw = *(p.eps ? p.eps : p.x);
    {
        // Both 16-byte pieces of a half row are requested unconditionally as well (HPW = 4, i.e. n <= 256: the first
        // piece again; steps >= 16 that would use it do not exist).  The conditional second piece compiled to
        // "load; s_waitcnt vmcnt(0)" AHEAD of the tile loads: every workgroup began by sitting out a memory round trip.
        const uint32_t* gbits = p.adj_bits + p.b_bits_off[b];
        const int second = HPW > 4 ? 1 : 0;
        const u32x4* ra = reinterpret_cast<const u32x4*>(gbits + (size_t)(min(rbA, W - 1) * 32 + i) * (2 * HPW) + h * HPW);
        const u32x4* rb = reinterpret_cast<const u32x4*>(gbits + (size_t)((two ? rbB : min(rbA, W - 1)) * 32 + i) * (2 * HPW) + h * HPW);
        // (12 bytes of the second piece: its last word would be steps 28-31, which no graph of <= 416 nodes has -- and a
        //  loaded register the compiler knows to be dead is reused at once, behind a vmcnt(0) for the write-after-write)
        const u32x4 a0 = ra[0], b0 = rb[0];
        const u32x3 a1 = *reinterpret_cast<const u32x3*>(ra + second), b1 = *reinterpret_cast<const u32x3*>(rb + second);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pkA[j] = a0[j]; pkB[j] = b0[j]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { pkA[4 + j] = a1[j]; pkB[4 + j] = b1[j]; }
        pkA[7] = 0u; pkB[7] = 0u;
    }
    GNM_MSTAMP(7)
    GNM_MSTAMP(1)
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int rq = ((wave + u * kAggmWaves) << 3) | rql;
        if (rq < (n16 >> 2)) {
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
    if (pro && p.p_gf) {       // readout partials: lanes with the same column chunk (lane bits 1-3 vary), then the waves
#pragma unroll
        for (int off = 2; off < 16; off <<= 1) {
            csum.x += __shfl_xor(csum.x, off, 64); csum.y += __shfl_xor(csum.y, off, 64);
            csum.z += __shfl_xor(csum.z, off, 64); csum.w += __shfl_xor(csum.w, off, 64);
        }
        if (rql == 0) rsum[wave * 8 + c4] = csum;
    }
    if constexpr (STATS) {      // (zeros where there is nothing: the epilogue multiplies without asking)
        if (tid < kAggmMaxN) dvs[tid] = (p.s_dsc1 && tid < n) ? dv_early : 0.f;
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
    const float selfB = p.self_loop ? 0.f : (p.eps ? 1.f + eps_raw : 1.f);
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
        const bool want_dot = STATS && p.deps_partial && !p.hfwd && !p.self_loop;
        const float cself = p.self_loop ? 1.f : selfB;
        const unsigned ybytes = (unsigned)(((size_t)(n - 1) * p.ldy + p.F) * 4);
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)row0 * p.ldy, 0, (int)ybytes, 0x00020000);
        // quarter blocks: 4 rows per lane at a time (r = 4 k + q: rows rb * 32 + 8 k + 4 h + q); two quarters' operands
        // are in flight (with 8-row halves the two operand sets spilled at 128 registers)
        struct Ops { float zr[4], xs[4]; int d0[4], d1[4]; };
        auto vrow_of = [&](int rb, int k, int q) { return rb * 32 + 8 * k + 4 * h + q; };
        // Addresses of a quarter's four elements (round 4): byte offset of the lane's element of row 4 h in block row 0
        // (computed once) + a wave-uniform quarter offset + q row strides -- one VALU add per access instead of a
        // clamp, a 64-bit multiply-add and a shift; rows past n are clipped by the buffer descriptors (loads return 0,
        // stores are dropped), a NARROW lane past F sits 2 GiB out.
        const unsigned row_y = (unsigned)p.ldy * 4u;
        const unsigned lane_y = (NARROW && col >= p.F) ? 0x80000000u : (unsigned)((4 * h * p.ldy + col) * 4);
        const unsigned row_z = STATS ? (unsigned)p.ldsz * 4u : 0u;
        const unsigned lane_z = STATS ? (unsigned)((4 * h * p.ldsz + col) * 4) : 0u;
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(STATS ? p.sZ + (size_t)row0 * p.ldsz : p.x), 0,
            STATS ? (int)(((size_t)(n - 1) * p.ldsz + p.F) * 4) : 0, 0x00020000);
        auto request = [&](int rb, int k, Ops& o) {
            const unsigned zq = lane_z + (unsigned)(rb * 32 + 8 * k) * row_z;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int vc = min(vrow_of(rb, k, q), n - 1);
                if constexpr (STATS) {
                    o.zr[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, zq + q * row_z, 0, 0));
                }
                if constexpr (AVG) {
                    o.xs[q] = p.x[(size_t)(row0 + vc) * p.ldx + (NARROW ? min(col, p.F - 1) : col)];
                    if constexpr (!STATS) { o.d0[q] = frp[vc]; o.d1[q] = frp[vc + 1]; }   // (forward only: the degree division)
                }
            }
        };
        auto finish = [&](int rb, int k, const f32x16& acc, const Ops& o) {
            float ex[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) ex[q] = 0.f;
            // The tile's own values of this lane's four elements, as phase A formed them (the three planes add up to
            // them exactly): rows 8 k' + 4 h .. + 3 of one column are 8 contiguous bytes of a plane's k-group, so ONE
            // ds_read_b64 per plane brings all four (round 4; until then twelve 2-byte reads per quarter, whose 16-byte
            // lane stride put four lanes on every bank: all of the kernel's SQ_LDS_BANK_CONFLICT and half of its
            // LDS-array cycles)
            f32x4 dv4 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (STATS) dv4 = *reinterpret_cast<const f32x4*>(dvs + rb * 32 + 8 * k + 4 * h);   // (rows >= n: unused)
            const unsigned yq = lane_y + (unsigned)(rb * 32 + 8 * k) * row_y;
            float wq[4];
            {
                const char* e = smem + (unsigned)min(rb * 4 + k, (n16 >> 3) - 1) * kAggmK8Stride + i * 16 + h * 8;
                const u32x2 p1 = *reinterpret_cast<const u32x2*>(e);
                const u32x2 p2 = *reinterpret_cast<const u32x2*>(e + plane_bytes);
                const u32x2 p3 = *reinterpret_cast<const u32x2*>(e + 2u * plane_bytes);
                wq[0] = (__uint_as_float(p1.x << 16) + __uint_as_float(p2.x << 16)) + __uint_as_float(p3.x << 16);
                wq[1] = (__uint_as_float(p1.x & 0xFFFF0000u) + __uint_as_float(p2.x & 0xFFFF0000u)) + __uint_as_float(p3.x & 0xFFFF0000u);
                wq[2] = (__uint_as_float(p1.y << 16) + __uint_as_float(p2.y << 16)) + __uint_as_float(p3.y << 16);
                wq[3] = (__uint_as_float(p1.y & 0xFFFF0000u) + __uint_as_float(p2.y & 0xFFFF0000u)) + __uint_as_float(p3.y & 0xFFFF0000u);
            }
            if constexpr (STATS) {
                if (shuffled) {        // workgroup-uniform and rare (the workgroups of the first B rows of the batch)
                    int gq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) gq[q] = gnm_perm_entry(p.s_inv_perm[min(row0 + min(vrow_of(rb, k, q), n - 1), p.n_batch - 1)], p.n_batch);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = p.s_s2sum[gq[q]] * p.s_U[(size_t)gq[q] * p.ld_U + col];
                        ex[q] = row0 + min(vrow_of(rb, k, q), n - 1) < p.n_batch ? t : 0.f;
                    }
                }
            }
            // (values of k-groups past the graph's last: the read above was clamped to a valid group -- drop them)
            const bool in_tile = rb * 4 + k < (n16 >> 3);            // wave-uniform
            float dq = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int vrow = vrow_of(rb, k, q);
                const float wv = in_tile ? wq[q] : 0.f;
                float tot;
                float sb = wv;
                if constexpr (AVG) {
                    tot = acc[4 * k + q];
                    if (p.self_loop) tot += wv;
                    if constexpr (!STATS) {
                        if (need_deg) tot /= (float)(o.d1[q] - o.d0[q] + p.self_loop);  // 0/0 -> NaN as in the reference
                    }
                    sb = need_xs ? o.xs[q] : wv;
                    if (!p.self_loop) tot += selfB * sb;
                    if (vrow >= n) sb = 0.f;
                } else {
                    tot = fmaf(cself, wv, acc[4 * k + q]);       // self loop: + h; eps form: + (1 + eps) h
                }
                if constexpr (STATS) {
                    const float zrow = o.zr[q];
                    const float pre = fmaf(zrow, lsc, lsh);
                    dq = fmaf(sb, gnm_relu(pre), dq);              // h as the forward formed it (rows >= n: sb = 0)
                    tot += s_pb + dv4[q] * s_ub;
                    tot += ex[q];
                    // the ReLU mask of the layer below; rows past n (their loads returned 0) count as masked, so the
                    // column sums need no guard of their own (the store of such a row is clipped anyway)
                    if (!(pre > 0.f && vrow < n)) tot = 0.f;
                    ss1 += tot;
                    ss2 = fmaf(tot, zrow - lmu, ss2);
                }
                // (row offset in the vector operand, scalar offset 0: see linear.hip, gnm_lin_stream_kernel)
                // (NARROW: a lane whose column is past F stores to an offset the descriptor clips)
                // (16-byte stores after a 4 x 4 DPP transpose inside lane quads -- 4 instead of 16 store instructions
                //  per row block -- were measured in round 4: +7 us per launch, the 16 extra VALU per quarter cost more
                //  than the store issue they save)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tot), ry, yq + q * row_y, 0, 0);
            }
            if constexpr (STATS) {
                if (want_dot) dot += (double)dq;      // (four fp32 products summed in fp32, the partials in fp64)
            }
        };
        Ops oa, ob;
        request(rbA, 0, oa);       // the first quarter's operands travel under the product
        GNM_MSTAMP(3)
        if (two) product(std::true_type{});
        else product(std::false_type{});
        // (s_setprio 2 around the product: +1 us, round 4)
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

// ---- persistent forward form (round 4; OPT-IN: GNM_AGGM_PERSIST=1) -----------------------------------------------
// An experiment kept for its measurements; the product launches gnm_aggm_kernel.  The kernel above spends a third of a
// workgroup's life on its tile load (the ~11 B/clk a CU keeps in flight) and LDS lets only two workgroups per CU overlap
// their phases.  This form keeps two workgroups per CU RESIDENT and walks each over its units (graph x 32-column block):
//
//   B1 | product (all steps) | self term of both row blocks (from the planes)
//      | the NEXT unit's tile (two items = eight 16-byte loads per lane) and bit rows requested
//      | epilogue: 16 stores per row block, no loads, no LDS | B0: the planes are dead
//      | the next unit's tile split into the planes | B1 ...
//
// The next tile travels under the epilogue's stores and the wait for the workgroup's slowest wave; nothing is in flight
// during a product, so the product keeps the register budget it has in gnm_aggm_kernel (116 registers, no scratch).
// Measured (B = 1024 x n = 400, F = 64, paired with gnm_aggm_kernel in one process): 76.6 us against 70.2 us fused, 77
// against 69 plain.  What it took to get there, for whoever picks this up:
//   * a first version split the product at row 256 and kept one item in flight through both halves: 48-72 spilled
//     registers at the 128 of four waves per SIMD, the reloads (each an s_waitcnt vmcnt(0)) inside the product;
//   * the unit descriptors have to come through s_load by hand (unit_words below): 84 us with vector loads;
//   * every wave issues the same number of loads and stores per round, branch-free, or the tile is waited for with
//     vmcnt(0) -- behind the wave's own stores;
//   * delaying the CU's second workgroup (HW_ID.TG_ID & 1) by k x 3.6 us so that one multiplies while the other loads:
//     k = 0..4 -> 76.6, 78, 83, 87, 88 us.  The tile (51 KB a unit) still arrives AFTER the product that hid nothing;
//     hiding it needs the tile in flight during the product, i.e. 32 more registers or LDS this kernel does not have.
// Equal units in lockstep lose to the 2048 short workgroups of gnm_aggm_kernel, whose start times the dispatcher
// staggers by itself.  Forward "sum" forms only (plain and fused BatchNorm + ReLU + readout prologue).
__global__ void __launch_bounds__(kAggmThreads, 4) gnm_aggp_kernel(const AggArgs p, const int units_total, const int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // GNM_AGGP_STAGGER: the workgroup in a CU's second slot starts a fraction of a unit late (persistent workgroups with
    // equal units otherwise run in lockstep: both workgroups of a CU load, multiply and store at the same time)
    if (__builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4) & 1)      // HW_ID.TG_ID: the workgroup's slot in its CU
        for (int k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(127);
    const int nc = p.F >> 5;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const unsigned plane_bytes = (unsigned)(p.n16_max >> 3) * kAggmK8Stride;
    char* lut = smem + 3u * plane_bytes;
    float4* rsum = reinterpret_cast<float4*>(lut + 128);
    const bool pro = p.p_scale != nullptr;
    const float eps_raw = *(p.eps ? p.eps : p.x);
    const float cself = p.self_loop ? 1.f : (p.eps ? 1.f + eps_raw : 1.f);
    if (tid < 16) {          // nibble e -> bf16 (bit 0, bit 1, bit 2, bit 3) as two words
        const unsigned one = 0x3F80u;
        u32x2 v;
        v.x = ((tid & 1) ? one : 0u) | ((tid & 2) ? one << 16 : 0u);
        v.y = ((tid & 4) ? one : 0u) | ((tid & 8) ? one << 16 : 0u);
        *reinterpret_cast<u32x2*>(lut + 8 * tid) = v;
    }
    // Per-lane constants of the load / split / store phases are re-derived inside the unit loop from a lane number the
    // compiler cannot see through (an empty asm): hoisted out of the loop they were ~36 registers of offsets kept alive
    // through the products, and the allocator answered by spilling the bit rows INSIDE the product.
    auto opaque_lane = [&]() -> int { int L = lane; asm volatile("" : "+v"(L)); return L; };

    // A unit's descriptor words (node_off[b], node_off[b + 1], b_bits_off[b]) through the SCALAR cache, by hand: inside
    // the unit loop the compiler may not use s_load itself (the loop stores to memory, and nothing tells it that those
    // stores never hit the descriptors), and as vector loads each of them was a full s_waitcnt vmcnt(0) -- three memory
    // latencies per unit, the second one draining the tile loads issued just before it.
    auto unit_words = [&](int b, int& row0, int& n, long long& boff) {
        unsigned long long w01, wb;
        const int32_t* pn = p.node_off + b;
        const int64_t* pb = p.b_bits_off + b;
        asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(w01), "=&s"(wb) : "s"(pn), "s"(pb) : "memory");
        row0 = (int)(unsigned)w01; n = (int)(unsigned)(w01 >> 32) - row0; boff = (long long)wb;
    };
    // first unit >= `unit` (stride: the grid) that has rows; graphs without nodes get their (empty) readout here
    auto next_valid = [&](int unit, int& b, int& cb, int& row0, int& n, long long& boff) -> int {
        for (; unit < units_total; unit += (int)gridDim.x) {
            const int grp = unit / (8 * nc), within = unit - grp * (8 * nc);
            b = grp * 8 + (within & 7); cb = within >> 3;
            if (b >= p.n_graphs) continue;
            unit_words(b, row0, n, boff);
            if (n > 0) return unit;
            if (pro && p.p_gf && tid < 32) p.p_gf[(size_t)b * p.p_ldgf + cb * 32 + tid] = p.p_gf_avg ? 0.f / 0.f : 0.f;
        }
        return unit;
    };
    // (every per-unit scalar below goes through readfirstlane: they are loop-carried through a loop that contains lane-
    //  dependent branches, the compiler's divergence analysis then keeps them -- and every address, bound and descriptor
    //  derived from them -- in vector registers: 174 registers and waterfall loops in the first build of this kernel)
    // the tile of the unit at (row0, n, col0): item u (0: rows 0-255, 1: rows 256..) = four rows x four columns per lane,
    // through a buffer descriptor (rows past n read zeros) with 32-bit per-lane offsets
    float4 v[2][4];
    const unsigned row_x = (unsigned)p.ldx * 4u;
    auto load_tile = [&](int L, int row0, int n, int col0) {       // n == 0: a descriptor of no bytes, every load returns zeros
        const int c4 = (L & 1) | ((L >> 4) << 1), rql = (L >> 1) & 7;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x) + (size_t)row0 * p.ldx + col0, 0, n > 0 ? (int)((((size_t)(n - 1) * p.ldx) + 32) * 4) : 0, 0x00020000);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned lx = (unsigned)((((wave + u * kAggmWaves) << 3) | rql) * 4) * row_x + 16u * (unsigned)c4;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                v[u][r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, lx + r * row_x, 0, 0));
        }
    };
    // the prologue's per-column scale and shift of a unit: requested AHEAD of its tile (behind the epilogue's stores they
    // would be the youngest operation in flight, and the split would wait for every store before it)
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_pro = [&](int L, int col0) {
        if (pro) {
            const int c4 = (L & 1) | ((L >> 4) << 1);
            psc = *reinterpret_cast<const float4*>(p.p_scale + col0 + 4 * c4);
            psh = *reinterpret_cast<const float4*>(p.p_shift + col0 + 4 * c4);
        }
    };
    auto split_tile = [&](int L, int row0, int n, int n16, int col0) {
        const int c4 = (L & 1) | ((L >> 4) << 1), rql = (L >> 1) & 7;
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rq = ((wave + u * kAggmWaves) << 3) | rql;
            if (rq < (n16 >> 2)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * rq + r;
                    float4 w = row < n ? v[u][r] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pro && row < n) {
                        w.x = gnm_relu(w.x * psc.x + psh.x); w.y = gnm_relu(w.y * psc.y + psh.y);
                        w.z = gnm_relu(w.z * psc.z + psh.z); w.w = gnm_relu(w.w * psc.w + psh.w);
                        if (p.p_hout) *reinterpret_cast<float4*>(p.p_hout + (size_t)(row0 + row) * p.p_ldh + col0 + 4 * c4) = w;
                        csum.x += w.x; csum.y += w.y; csum.z += w.z; csum.w += w.w;
                    }
                    v[u][r] = w;
                }
                const unsigned base = (unsigned)(rq >> 1) * kAggmK8Stride + (unsigned)((4 * c4 * 8 + 4 * (rq & 1)) * 2);
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
        if (pro && p.p_gf) {      // readout partials: lanes with the same column chunk (lane bits 1-3 vary), then the waves
#pragma unroll
            for (int off = 2; off < 16; off <<= 1) {
                csum.x += __shfl_xor(csum.x, off, 64); csum.y += __shfl_xor(csum.y, off, 64);
                csum.z += __shfl_xor(csum.z, off, 64); csum.w += __shfl_xor(csum.w, off, 64);
            }
            if (rql == 0) rsum[wave * 8 + c4] = csum;
        }
    };
    unsigned pkA[8], pkB[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { pkA[j] = 0u; pkB[j] = 0u; }
    // this wave's bit rows of graph b (row blocks rbA, rbB; layout: gnm_aggm_kernel), in two requests: the first 16-byte
    // piece of each half row (steps 0-15) ahead of the product, the second (steps 16-27, 12 bytes) from INSIDE it, at
    // step 6 -- six registers that would otherwise sit unused through the first sixteen steps, which the allocator
    // took as its spill candidates (reloaded from scratch at every step of the second half)
    auto bit_rows = [&](int L, long long boff, int W, int rbA, int rbB, bool two, const u32x4*& ra, const u32x4*& rb) -> int {
        const int HPW = (((W + 1) >> 1) + 3) & ~3;
        const uint32_t* gbits = p.adj_bits + boff;
        const int ii = L & 31, hh = L >> 5;
        ra = reinterpret_cast<const u32x4*>(gbits + (size_t)(min(rbA, W - 1) * 32 + ii) * (2 * HPW) + hh * HPW);
        rb = reinterpret_cast<const u32x4*>(gbits + (size_t)((two ? rbB : min(rbA, W - 1)) * 32 + ii) * (2 * HPW) + hh * HPW);
        return HPW > 4 ? 1 : 0;
    };
    auto load_bits_lo = [&](long long b, int W, int rbA, int rbB, bool two) {
        const u32x4 *ra, *rb;
        bit_rows(opaque_lane(), b, W, rbA, rbB, two, ra, rb);
        const u32x4 a0 = ra[0], b0 = rb[0];
#pragma unroll
        for (int j = 0; j < 4; ++j) { pkA[j] = a0[j]; pkB[j] = b0[j]; }
    };
    auto load_bits_hi = [&](long long b, int W, int rbA, int rbB, bool two) {
        const u32x4 *ra, *rb;
        const int second = bit_rows(opaque_lane(), b, W, rbA, rbB, two, ra, rb);
        const u32x3 a1 = *reinterpret_cast<const u32x3*>(ra + second), b1 = *reinterpret_cast<const u32x3*>(rb + second);
#pragma unroll
        for (int j = 0; j < 3; ++j) { pkA[4 + j] = a1[j]; pkB[4 + j] = b1[j]; }
    };
    const char* bp0 = smem + h * kAggmK8Stride + i * 16;
    const char* bp1 = bp0 + plane_bytes;
    const char* bp2 = bp1 + plane_bytes;
    f32x16 accA, accB;
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
    auto product = [&](auto two_tag, int ksteps, long long b, int W, int rbA, int rbB) {   // software-pipelined as in gnm_aggm_kernel
        constexpr bool TWO = decltype(two_tag)::value;
        bf16x8 b0 = bfrag(bp0, 0), b1 = bfrag(bp1, 0), b2 = bfrag(bp2, 0);
        bf16x8 aA = afrag(pkA[0], 0), aB = aA;
        if constexpr (TWO) aB = afrag(pkB[0], 0);
#pragma unroll
        for (int ks = 0; ks < 26; ++ks) {
            if (ks < ksteps) {                                // wave-uniform
                if (ks == 6 && ksteps > 16) load_bits_hi(b, W, rbA, rbB, TWO);    // (steps 16.. are its only readers)
                constexpr int LASTK = 25;
                const int kn = ks < LASTK ? ks + 1 : LASTK;
                const bf16x8 n0 = bfrag(bp0, kn), n1 = bfrag(bp1, kn), n2 = bfrag(bp2, kn);
                const bf16x8 nA = afrag(pkA[kn >> 2], kn & 3);
                bf16x8 nB = nA;
                if constexpr (TWO) nB = afrag(pkB[kn >> 2], kn & 3);
                __builtin_amdgcn_sched_barrier(0);
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
    // acc += cself x (the tile's own values of row block rb): the planes add up to them exactly
    auto selfadd = [&](f32x16& acc, int rb, int n16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const char* e = smem + (unsigned)min(rb * 4 + k, (n16 >> 3) - 1) * kAggmK8Stride + i * 16 + h * 8;
            const u32x2 p1 = *reinterpret_cast<const u32x2*>(e);
            const u32x2 p2 = *reinterpret_cast<const u32x2*>(e + plane_bytes);
            const u32x2 p3 = *reinterpret_cast<const u32x2*>(e + 2u * plane_bytes);
            const bool in_tile = rb * 4 + k < (n16 >> 3);
            const float w0 = (__uint_as_float(p1.x << 16) + __uint_as_float(p2.x << 16)) + __uint_as_float(p3.x << 16);
            const float w1 = (__uint_as_float(p1.x & 0xFFFF0000u) + __uint_as_float(p2.x & 0xFFFF0000u)) + __uint_as_float(p3.x & 0xFFFF0000u);
            const float w2 = (__uint_as_float(p1.y << 16) + __uint_as_float(p2.y << 16)) + __uint_as_float(p3.y << 16);
            const float w3 = (__uint_as_float(p1.y & 0xFFFF0000u) + __uint_as_float(p2.y & 0xFFFF0000u)) + __uint_as_float(p3.y & 0xFFFF0000u);
            acc[4 * k + 0] = fmaf(cself, in_tile ? w0 : 0.f, acc[4 * k + 0]);
            acc[4 * k + 1] = fmaf(cself, in_tile ? w1 : 0.f, acc[4 * k + 1]);
            acc[4 * k + 2] = fmaf(cself, in_tile ? w2 : 0.f, acc[4 * k + 2]);
            acc[4 * k + 3] = fmaf(cself, in_tile ? w3 : 0.f, acc[4 * k + 3]);
        }
    };

    // ---- the first unit ---------------------------------------------------------------------------------------------
    int b, cb, row0, n;
    long long boff;
    int unit = next_valid((int)blockIdx.x, b, cb, row0, n, boff);
    if (unit >= units_total) return;
    int col0 = cb * 32;
    load_pro(opaque_lane(), col0);
    load_tile(opaque_lane(), row0, n, col0);
    {
        const int W = (n + 31) >> 5, role = (wave + cb) % kAggmWaves;
        __builtin_amdgcn_sched_barrier(0);
        load_bits_lo(boff, W, role, role + kAggmWaves, role + kAggmWaves < W);
    }
    split_tile(opaque_lane(), row0, n, ((n + 15) >> 4) * 16, col0);
    __syncthreads();                                                             // B1 of the first unit

    while (true) {                                                               // workgroup-uniform
        const int W = (n + 31) >> 5;
        const int ksteps = (n + 15) >> 4;
        const int n16 = ksteps * 16;
        const int role = (wave + cb) % kAggmWaves;
        const int rbA = role, rbB = role + kAggmWaves;
        const bool two = rbB < W;
        const bool has_rows = rbA < W;
        if (pro && p.p_gf && tid < 8) {      // (the shares were written before B1; the next ones come after B0)
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int w = 0; w < kAggmWaves; ++w) {
                const float4 s = rsum[w * 8 + tid];
                t.x += s.x; t.y += s.y; t.z += s.z; t.w += s.w;
            }
            if (p.p_gf_avg) {
                const float inv = 1.f / (float)n;
                t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
            }
            *reinterpret_cast<float4*>(p.p_gf + (size_t)b * p.p_ldgf + col0 + 4 * tid) = t;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
        __builtin_amdgcn_sched_barrier(0);
        if (has_rows) {
            if (two) {
                product(std::true_type{}, ksteps, boff, W, rbA, rbB);
                __builtin_amdgcn_sched_barrier(0);
                selfadd(accB, rbB, n16);
            } else {
                product(std::false_type{}, ksteps, boff, W, rbA, rbB);
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(pkB[j]));
            }
            __builtin_amdgcn_sched_barrier(0);
            selfadd(accA, rbA, n16);
        } else {
            // (a wave "uses" the bit rows it requested on the paths that do not need them, here and above: left pending,
            //  they made the compiler guard the next round's first writes to those registers with a wait that the
            //  scale/shift loads -- a memory latency in front of the epilogue's stores -- had to satisfy)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(pkA[j]), "v"(pkB[j]));
        }
        __builtin_amdgcn_sched_barrier(0);
        // The next unit: its tile and bit rows travel under the epilogue and the wait for the slowest wave.  Both the
        // loads and the stores below are issued by EVERY wave in EVERY round (a workgroup without a next unit loads
        // through an empty descriptor, a wave without a row block stores past the end of its descriptor: the hardware
        // drops both), so that the number of memory operations between a tile load and its use does not depend on a
        // branch -- with the branches the compiler had to wait for the tile with s_waitcnt vmcnt(0), i.e. until this
        // wave's own stores had drained as well.
        int nb = b, ncb = cb, nrow0 = row0, nn = 0;
        long long nboff = boff;
        const int nunit = next_valid(unit + (int)gridDim.x, nb, ncb, nrow0, nn, nboff);
        const bool more = nunit < units_total;
        if (!more) nn = 0;
        const int ncol0 = ncb * 32;
        if (ncb != cb) load_pro(opaque_lane(), ncol0);        // (the grid is a multiple of 8 nc for every nc that divides 64: rare)
        load_tile(opaque_lane(), nrow0, nn, ncol0);
        {
            const int nW = more ? (nn + 31) >> 5 : W, nrole = (wave + ncb) % kAggmWaves;
            load_bits_lo(nboff, nW, nrole, nrole + kAggmWaves, nrole + kAggmWaves < nW);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned ybytes = (unsigned)(((size_t)(n - 1) * p.ldy + p.F) * 4);
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)row0 * p.ldy, 0, (int)ybytes, 0x00020000);
            const unsigned row_y = (unsigned)p.ldy * 4u;
            const int Ly = opaque_lane();
            const unsigned lane_y = (unsigned)((4 * (Ly >> 5) * p.ldy + col0 + (Ly & 31)) * 4);
            const unsigned yA = has_rows ? lane_y + (unsigned)(rbA * 32) * row_y : 0x80000000u;
            const unsigned yB = two ? lane_y + (unsigned)(rbB * 32) * row_y : 0x80000000u;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accA[4 * k + q]), ry, yA + (unsigned)(8 * k + q) * row_y, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accB[4 * k + q]), ry, yB + (unsigned)(8 * k + q) * row_y, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!more) break;
        unit = nunit; b = nb; cb = ncb; row0 = nrow0; n = nn; col0 = ncol0; boff = nboff;
        __syncthreads();                                                         // B0: every wave is through with the planes
        __builtin_amdgcn_sched_barrier(0);
        split_tile(opaque_lane(), row0, n, ((n + 15) >> 4) * 16, col0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                                         // B1
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

static bool aggm_persist() { static const bool v = gnm_env_int("GNM_AGGM_PERSIST", 0) != 0; return v; }
static int aggm_cu_count() {          // per device, asked once
    static int cus[kGnmMaxDevices] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kGnmMaxDevices) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
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
#define GNM_AGGM_LAUNCH(S_, A_)                                                                              \
    do {                                                                                                     \
        GNM_ALLOW_FULL_LDS((&gnm_aggm_kernel<S_, A_>));                                                      \
        hipLaunchKernelGGL((gnm_aggm_kernel<S_, A_>), dim3(grid), dim3(kAggmThreads), lds, stream, a);       \
    } while (0)
    // opt-in experiment (GNM_AGGM_PERSIST=1; slower than the launches below, see gnm_aggp_kernel): forward "sum" forms
    // on two resident workgroups per CU walking the units (one per CU where a plane set is too large for two)
    if (aggm_persist() && !stats && !a.average && !a.backward && !a.deps_partial && a.F >= 32 && a.y) {
        const int per_cu = 2 * lds <= (size_t)kLdsBudget ? 2 : 1;
        int pgrid = aggm_cu_count() * per_cu;
        if (pgrid > grid) pgrid = grid;
        GNM_ALLOW_FULL_LDS((&gnm_aggp_kernel));
        static const int stagger = gnm_env_int("GNM_AGGP_STAGGER", 0);      // x 3.6 us (s_sleep 127 at 2.25 GHz)
        hipLaunchKernelGGL(gnm_aggp_kernel, dim3(pgrid), dim3(kAggmThreads), lds, stream, a, grid, per_cu == 2 ? stagger : 0);
        GNM_CHECK_LAUNCH();
        return GNM_OK;
    }
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
