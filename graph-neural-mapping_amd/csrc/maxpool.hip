// neighbor_pooling_type == "max" (graphcnn.py:55-81 padded neighbour list, :137-143 maxpool, used at :149-151 /
// :173-175): element-wise maximum over a node's neighbour rows.
//
// The reference gathers h_with_dummy[padded_neighbor_list] ([N, max_deg (+1), F]) and takes torch.max over dim 1.  Here
// the padded list is never built: a row's candidates are, in the reference's order,
//     its neighbours (graph.neighbors[j] order), then ONE dummy candidate when its degree is below the batch's
//     max_deg (the padding slots all hold the same row, the column minimum of h: graphcnn.py:140), then the node
//     itself when learn_eps is False (:73-74),
// scanned with ATen's CPU rule for max over a dimension (a later candidate replaces the running one only if it is
// greater, or is the first NaN), so the selected index -- where autograd sends the gradient -- is the reference's.
// The backward is a gather over the transposed, de-duplicated neighbour structure (no atomics: bitwise repeatable).
//
// Not a roofline kernel: this mode is outside BASELINE.json's north_star (sum / average aggregation); rows are
// gathered straight from L2 / HBM, one lane per (row, column).
#include "gnm_common.h"

static constexpr int kMaxpoolThreads = 256;
static constexpr int kColminRows = 256;           // rows per workgroup of the column-minimum pass

// (value, row) candidates of a column minimum with torch.min's CPU semantics: the first NaN wins, else the smaller
// value, ties to the smaller row.  Associative, so any reduction tree gives the sequential scan's answer.
__device__ __forceinline__ bool colmin_better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v < bv || (v == bv && i < bi);
}

__global__ void __launch_bounds__(kMaxpoolThreads) gnm_colmin_partial_kernel(const float* __restrict__ h, int ldh, int N,
                                                                             int F, float* __restrict__ pval,
                                                                             int* __restrict__ pidx) {
    __shared__ float sv[kMaxpoolThreads];
    __shared__ int si[kMaxpoolThreads];
    const int lane_c = threadIdx.x & 63, phase = threadIdx.x >> 6;          // 64 columns x 4 row phases
    const int r0 = blockIdx.x * kColminRows;
    const int r1 = min(N, r0 + kColminRows);
    for (int c0 = 0; c0 < F; c0 += 64) {
        const int c = c0 + lane_c;
        float bv = 0.f;
        int bi = -1;
        if (c < F) {
            int r = r0 + phase;
            for (; r + 28 < r1; r += 32) {                      // 8 independent loads in flight
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = h[(size_t)(r + 4 * u) * ldh + c];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (bi < 0 || colmin_better(v[u], r + 4 * u, bv, bi)) { bv = v[u]; bi = r + 4 * u; }
            }
            for (; r < r1; r += 4) {
                const float v = h[(size_t)r * ldh + c];
                if (bi < 0 || colmin_better(v, r, bv, bi)) { bv = v; bi = r; }
            }
        }
        sv[threadIdx.x] = bv;
        si[threadIdx.x] = bi;
        __syncthreads();
        if (phase == 0 && c < F) {
            for (int p = 1; p < 4; ++p) {
                const float v = sv[p * 64 + lane_c];
                const int i = si[p * 64 + lane_c];
                if (i >= 0 && (bi < 0 || colmin_better(v, i, bv, bi))) { bv = v; bi = i; }
            }
            pval[(size_t)blockIdx.x * F + c] = bv;
            pidx[(size_t)blockIdx.x * F + c] = bi;
        }
        __syncthreads();
    }
}

// second stage: 64 columns x 16 phases of a workgroup walk the partials (8 loads in flight each), LDS combine
__global__ void __launch_bounds__(1024) gnm_colmin_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx,
                                                                int nblk, int F, float* __restrict__ vmin,
                                                                int* __restrict__ amin) {
    __shared__ float sv[1024];
    __shared__ int si[1024];
    const int lane_c = threadIdx.x & 63, phase = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane_c;
    float bv = 0.f;
    int bi = -1;
    if (c < F) {
        int b = phase;
        for (; b + 7 * 16 < nblk; b += 8 * 16) {
            float v[8];
            int id[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = pval[(size_t)(b + 16 * u) * F + c];
                id[u] = pidx[(size_t)(b + 16 * u) * F + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (id[u] >= 0 && (bi < 0 || colmin_better(v[u], id[u], bv, bi))) { bv = v[u]; bi = id[u]; }
        }
        for (; b < nblk; b += 16) {
            const float v = pval[(size_t)b * F + c];
            const int i = pidx[(size_t)b * F + c];
            if (i >= 0 && (bi < 0 || colmin_better(v, i, bv, bi))) { bv = v; bi = i; }
        }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    if (phase == 0 && c < F) {
        for (int p = 1; p < 16; ++p) {
            const float v = sv[p * 64 + lane_c];
            const int i = si[p * 64 + lane_c];
            if (i >= 0 && (bi < 0 || colmin_better(v, i, bv, bi))) { bv = v; bi = i; }
        }
        vmin[c] = bv;
        amin[c] = bi;
    }
}

// ATen's max over a dimension on the CPU (what the reference's torch.max(..., dim = 1) runs): `if (!(value <= max))`
// replaces the running maximum and a NaN ends the scan.
#define GNM_MAX_STEP(v_, j_)                                   \
    do {                                                       \
        const float v__ = (v_);                                \
        if (!have) { best = v__; idx = (j_); have = true; }    \
        else if (best == best && !(v__ <= best)) { best = v__; idx = (j_); } \
    } while (0)

// pooled + (1 + eps) * h (graphcnn.py:161) rounded operation by operation as torch evaluates it: no fused multiply-add
__device__ __forceinline__ float eps_form(float best, float eps, float hv) {
#pragma clang fp contract(off)
    const float t = (1.0f + eps) * hv;
    return best + t;
}

__global__ void __launch_bounds__(kMaxpoolThreads) gnm_maxpool_fwd_kernel(
    const float* __restrict__ h, int ldh, const int* __restrict__ nb_off, const int* __restrict__ nb_col, long long total,
    int F, int max_deg, int self_last, const float* __restrict__ eps, const float* __restrict__ dummy,
    float* __restrict__ out, int ldo, int* __restrict__ amax) {
    const long long t = (long long)blockIdx.x * kMaxpoolThreads + threadIdx.x;
    if (t >= total) return;
    const int i = (int)(t / F), c = (int)(t - (long long)i * F);
    const int lo = nb_off[i], hi = nb_off[i + 1];
    float best = 0.f;
    int idx = -2;
    bool have = false;
    for (int e = lo; e < hi; ++e) {
        const int j = nb_col[e];
        GNM_MAX_STEP(h[(size_t)j * ldh + c], j);
    }
    if (hi - lo < max_deg) GNM_MAX_STEP(dummy[c], -1);
    const float hv = h[(size_t)i * ldh + c];
    if (self_last) GNM_MAX_STEP(hv, i);
    float r = best;
    if (eps) r = eps_form(best, eps[0], hv);
    out[(size_t)i * ldo + c] = r;
    if (amax) amax[t] = idx;
}
#undef GNM_MAX_STEP

// d h[j] = sum over the rows i that list j (once per (i, j): t_off / t_col, ascending i) of dpooled[i] where i
// selected j, + (1 + eps) dpooled[j] for the eps form.  The self candidate of the self-loop form is a (j, j) pair of
// the structure.
__global__ void __launch_bounds__(kMaxpoolThreads) gnm_maxpool_bwd_kernel(
    const float* __restrict__ g, int ldg, const int* __restrict__ amax, const int* __restrict__ t_off,
    const int* __restrict__ t_col, long long total, int F, const float* __restrict__ eps, float* __restrict__ dh,
    int ldd) {
    const long long t = (long long)blockIdx.x * kMaxpoolThreads + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t / F), c = (int)(t - (long long)j * F);
    float acc = 0.f;
    for (int e = t_off[j]; e < t_off[j + 1]; ++e) {
        const int i = t_col[e];
        if (amax[(size_t)i * F + c] == j) acc += g[(size_t)i * ldg + c];
    }
    if (eps) acc = fmaf(1.0f + eps[0], g[(size_t)j * ldg + c], acc);      // (explicit: the tiled kernel must give the same bits)
    dh[(size_t)j * ldd + c] = acc;
}

// rows that selected the dummy (only rows without neighbours can: the column minimum never exceeds a neighbour's
// value): their gradient goes to the row torch.min picked for that column (graphcnn.py:140)
__global__ void gnm_maxpool_bwd_dummy_kernel(const float* __restrict__ g, int ldg, const int* __restrict__ amax,
                                             const int* __restrict__ iso_rows, int n_iso, const int* __restrict__ amin,
                                             int F, float* __restrict__ dh, int ldd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= F) return;
    float s = 0.f;
    for (int k = 0; k < n_iso; ++k) {
        const int i = iso_rows[k];
        if (amax[(size_t)i * F + c] == -1) s += g[(size_t)i * ldg + c];
    }
    dh[(size_t)amin[c] * ldd + c] += s;
}

// ---- tiled forms: one workgroup per graph, the graph's rows staged in LDS ------------------------------------------------
// Same candidates, same scan order, same results as the kernels above (tests compare them bitwise); what changes is
// where the neighbour rows come from.  A wave covers 64 / LPR rows at a time with LPR = F / 4 lanes x 16 B per row;
// a row's ids are fetched LPR at a time by its own lanes (one 4-byte load each, requested one chunk ahead) and handed
// round with DPP row broadcasts, so the inner step is one ds_read_b128 and four compare / select pairs.
static constexpr int kMaxTileThreads = 1024;
static constexpr int kMaxTileAhead = 4;          // id chunks requested ahead of the one being consumed

template <int LPR, int K>
__device__ __forceinline__ int bcast_id(int v) {
    if constexpr (LPR == 16) {
        return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, false);             // row_newbcast:K
    } else {                                                                              // 8-lane groups
        int r = __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0x3, false);
        return __builtin_amdgcn_update_dpp(r, v, 0x150 + 8 + K, 0xf, 0xc, false);
    }
}

// one candidate against the running maximum of the four columns a lane owns; ATen's rule (see GNM_MAX_STEP).  The
// running maximum starts at -inf with the FIRST candidate's row as its index, which is what the scan gives when that
// candidate is -inf itself and is overwritten by it otherwise; slots past the end of a row read a row of -inf, which
// never replaces anything (not even a NaN: the `best == best` term) -- so the loop needs no branches.
#define GNM_MAX4(v_, j_)                                                                 \
    do {                                                                                 \
        const float4 v__ = (v_);                                                         \
        const int j__ = (j_);                                                            \
        if (best.x == best.x && !(v__.x <= best.x)) { best.x = v__.x; ix.x = j__; }      \
        if (best.y == best.y && !(v__.y <= best.y)) { best.y = v__.y; ix.y = j__; }      \
        if (best.z == best.z && !(v__.z <= best.z)) { best.z = v__.z; ix.z = j__; }      \
        if (best.w == best.w && !(v__.w <= best.w)) { best.w = v__.w; ix.w = j__; }      \
    } while (0)

template <int LPR>
__global__ void __launch_bounds__(kMaxTileThreads) gnm_maxpool_tile_fwd_kernel(
    const float* __restrict__ h, int ldh, const int* __restrict__ nb_off, const int* __restrict__ nb_col,
    const int* __restrict__ node_off, int max_deg, int self_last, const float* __restrict__ eps,
    const float* __restrict__ dummy, float* __restrict__ out, int ldo, int* __restrict__ amax) {
    constexpr int F = 4 * LPR, RPW = 64 / LPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* tile = reinterpret_cast<float4*>(smem);
    const int b = blockIdx.x;
    const int row0 = node_off[b], n = node_off[b + 1] - row0;
    const int tid = threadIdx.x;
    {
        const int total = n * LPR;
        int i = tid;
        for (; i + 3 * kMaxTileThreads < total; i += 4 * kMaxTileThreads) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = i + u * kMaxTileThreads;
                v[u] = *reinterpret_cast<const float4*>(h + (size_t)(row0 + k / LPR) * ldh + 4 * (k % LPR));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) tile[i + u * kMaxTileThreads] = v[u];
        }
        for (; i < total; i += kMaxTileThreads)
            tile[i] = *reinterpret_cast<const float4*>(h + (size_t)(row0 + i / LPR) * ldh + 4 * (i % LPR));
        const float ninf = -__builtin_inff();
        if (tid < LPR) tile[total + tid] = make_float4(ninf, ninf, ninf, ninf);      // row n: what empty slots read
    }
    int* offs = reinterpret_cast<int*>(smem + (size_t)(n + 1) * F * 4);    // the graph's n + 1 list offsets
    for (int i = tid; i <= n; i += kMaxTileThreads) offs[i] = nb_off[row0 + i];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = lane % LPR, slot = lane / LPR;
    const float epsv = eps ? eps[0] : 0.f;
    float4 dm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dummy) dm = *reinterpret_cast<const float4*>(dummy + 4 * sub);
    const int gap = n + row0;                  // (empty slot: -1) -> local row n after the subtraction below
    for (int r0 = wave * RPW; r0 < n; r0 += (kMaxTileThreads / 64) * RPW) {             // wave-uniform
        const int i = r0 + slot;
        const bool valid = i < n;
        int lo = 0, hi = 0;
        if (valid) { lo = offs[i]; hi = offs[i + 1]; }
        int e0 = lo;
        int q[kMaxTileAhead];
#pragma unroll
        for (int a = 0; a < kMaxTileAhead; ++a) q[a] = (e0 + a * LPR + sub < hi) ? nb_col[e0 + a * LPR + sub] : -1;
        const float ninf = -__builtin_inff();
        float4 best = make_float4(ninf, ninf, ninf, ninf);
        // the first candidate: the first neighbour, else the dummy (a padded row), else the node itself
        int first = bcast_id<LPR, 0>(q[0]);
        if (first < 0) first = (hi - lo < max_deg) ? -1 : row0 + i;
        int4 ix = make_int4(first, first, first, first);
        while (__any(e0 < hi)) {
            const int cur = q[0];
#pragma unroll
            for (int a = 0; a + 1 < kMaxTileAhead; ++a) q[a] = q[a + 1];
            q[kMaxTileAhead - 1] = (e0 + kMaxTileAhead * LPR + sub < hi) ? nb_col[e0 + kMaxTileAhead * LPR + sub] : -1;
            e0 += LPR;
#define GNM_MAX_Q(K)                                                                                        \
    {                                                                                                       \
        const int j0 = bcast_id<LPR, K>(cur), j1 = bcast_id<LPR, K + 1>(cur);                               \
        const int j2 = bcast_id<LPR, K + 2>(cur), j3 = bcast_id<LPR, K + 3>(cur);                           \
        const float4 v0 = tile[((j0 < 0 ? gap : j0) - row0) * LPR + sub];                                   \
        const float4 v1 = tile[((j1 < 0 ? gap : j1) - row0) * LPR + sub];                                   \
        const float4 v2 = tile[((j2 < 0 ? gap : j2) - row0) * LPR + sub];                                   \
        const float4 v3 = tile[((j3 < 0 ? gap : j3) - row0) * LPR + sub];                                   \
        GNM_MAX4(v0, j0); GNM_MAX4(v1, j1); GNM_MAX4(v2, j2); GNM_MAX4(v3, j3);                             \
    }
            GNM_MAX_Q(0) GNM_MAX_Q(4)
            if constexpr (LPR == 16) { GNM_MAX_Q(8) GNM_MAX_Q(12) }
#undef GNM_MAX_Q
        }
        if (!valid) continue;
        if (hi - lo < max_deg) GNM_MAX4(dm, -1);
        const float4 hv = tile[i * LPR + sub];
        if (self_last) GNM_MAX4(hv, row0 + i);
        float4 r = best;
        if (eps) {
            r.x = eps_form(best.x, epsv, hv.x); r.y = eps_form(best.y, epsv, hv.y);
            r.z = eps_form(best.z, epsv, hv.z); r.w = eps_form(best.w, epsv, hv.w);
        }
        *reinterpret_cast<float4*>(out + (size_t)(row0 + i) * ldo + 4 * sub) = r;
        if (amax) *reinterpret_cast<int4*>(amax + (size_t)(row0 + i) * F + 4 * sub) = ix;
    }
}
#undef GNM_MAX4

typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

template <int LPR>
__global__ void __launch_bounds__(kMaxTileThreads) gnm_maxpool_tile_bwd_kernel(
    const float* __restrict__ g, int ldg, const int* __restrict__ amax, const int* __restrict__ t_off,
    const int* __restrict__ t_col, const int* __restrict__ node_off, const float* __restrict__ eps,
    float* __restrict__ dh, int ldd) {
    constexpr int F = 4 * LPR, RPW = 64 / LPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int row0 = node_off[b], n = node_off[b + 1] - row0;
    float4* gt = reinterpret_cast<float4*>(smem);                               // gradient rows of the graph (+ a zero row)
    u16x4* am = reinterpret_cast<u16x4*>(smem + (size_t)(n + 1) * F * 4);       // selected row, graph-local (0xFFFF: none here)
    const int tid = threadIdx.x;
    for (int i = tid; i < n * LPR; i += kMaxTileThreads) {
        const int r = i / LPR, c = i % LPR;
        gt[i] = *reinterpret_cast<const float4*>(g + (size_t)(row0 + r) * ldg + 4 * c);
        const int4 a = *reinterpret_cast<const int4*>(amax + (size_t)(row0 + r) * F + 4 * c);
        u16x4 l;
        l[0] = (unsigned short)(a.x < 0 ? 0xFFFF : a.x - row0); l[1] = (unsigned short)(a.y < 0 ? 0xFFFF : a.y - row0);
        l[2] = (unsigned short)(a.z < 0 ? 0xFFFF : a.z - row0); l[3] = (unsigned short)(a.w < 0 ? 0xFFFF : a.w - row0);
        am[i] = l;
    }
    if (tid < LPR) {                                   // row n: what empty slots read -- selects nobody
        gt[n * LPR + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        u16x4 none;
        none[0] = none[1] = none[2] = none[3] = 0xFFFF;
        am[n * LPR + tid] = none;
    }
    int* offs = reinterpret_cast<int*>(smem + (size_t)(n + 1) * F * 6);
    for (int i = tid; i <= n; i += kMaxTileThreads) offs[i] = t_off[row0 + i];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = lane % LPR, slot = lane / LPR;
    const int gap = n + row0;
    for (int r0 = wave * RPW; r0 < n; r0 += (kMaxTileThreads / 64) * RPW) {
        const int j = r0 + slot;
        const bool valid = j < n;
        int lo = 0, hi = 0;
        if (valid) { lo = offs[j]; hi = offs[j + 1]; }
        const unsigned short jl = (unsigned short)j;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int e0 = lo;
        int q[kMaxTileAhead];
#pragma unroll
        for (int a = 0; a < kMaxTileAhead; ++a) q[a] = (e0 + a * LPR + sub < hi) ? t_col[e0 + a * LPR + sub] : -1;
        while (__any(e0 < hi)) {
            const int cur = q[0];
#pragma unroll
            for (int a = 0; a + 1 < kMaxTileAhead; ++a) q[a] = q[a + 1];
            q[kMaxTileAhead - 1] = (e0 + kMaxTileAhead * LPR + sub < hi) ? t_col[e0 + kMaxTileAhead * LPR + sub] : -1;
            e0 += LPR;
            // (adding the 0.f of a row that did not select j leaves the sum's bits as skipping it would)
#define GNM_BWD_1(a_, g_)                                    \
    acc.x += (a_[0] == jl) ? g_.x : 0.f;                     \
    acc.y += (a_[1] == jl) ? g_.y : 0.f;                     \
    acc.z += (a_[2] == jl) ? g_.z : 0.f;                     \
    acc.w += (a_[3] == jl) ? g_.w : 0.f;
#define GNM_BWD_Q(K)                                                                                        \
    {                                                                                                       \
        const int i0 = bcast_id<LPR, K>(cur), i1 = bcast_id<LPR, K + 1>(cur);                               \
        const int i2 = bcast_id<LPR, K + 2>(cur), i3 = bcast_id<LPR, K + 3>(cur);                           \
        const int l0 = ((i0 < 0 ? gap : i0) - row0) * LPR + sub, l1 = ((i1 < 0 ? gap : i1) - row0) * LPR + sub; \
        const int l2 = ((i2 < 0 ? gap : i2) - row0) * LPR + sub, l3 = ((i3 < 0 ? gap : i3) - row0) * LPR + sub; \
        const u16x4 a0 = am[l0], a1 = am[l1], a2 = am[l2], a3 = am[l3];                                     \
        const float4 g0 = gt[l0], g1 = gt[l1], g2 = gt[l2], g3 = gt[l3];                                    \
        GNM_BWD_1(a0, g0) GNM_BWD_1(a1, g1) GNM_BWD_1(a2, g2) GNM_BWD_1(a3, g3)                             \
    }
            GNM_BWD_Q(0) GNM_BWD_Q(4)
            if constexpr (LPR == 16) { GNM_BWD_Q(8) GNM_BWD_Q(12) }
#undef GNM_BWD_Q
#undef GNM_BWD_1
        }
        if (!valid) continue;
        if (eps) {
            const float s = 1.0f + eps[0];
            const float4 gv = gt[j * LPR + sub];
            acc.x = fmaf(s, gv.x, acc.x); acc.y = fmaf(s, gv.y, acc.y);
            acc.z = fmaf(s, gv.z, acc.z); acc.w = fmaf(s, gv.w, acc.w);
        }
        *reinterpret_cast<float4*>(dh + (size_t)(row0 + j) * ldd + 4 * sub) = acc;
    }
}

static bool maxpool_tile_ok(const void* a, int lda, const void* b, int ldb, int F, int n_max, size_t lds) {
    return (F == 32 || F == 64) && n_max > 0 && n_max < 65535 && lds <= (size_t)kLdsBudget && (lda & 3) == 0 &&
           (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0;
}

extern "C" int gnm_maxpool_colmin_blocks(int N) { return N > 0 ? (N + kColminRows - 1) / kColminRows : 1; }

extern "C" int gnm_maxpool_colmin(const float* h, int ldh, int N, int F, float* ws_val, int* ws_idx, float* vmin,
                                  int* amin, void* stream) {
    if (N <= 0 || F <= 0 || !h || !ws_val || !ws_idx || !vmin || !amin || ldh < F) return GNM_ERR_BAD_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nblk = gnm_maxpool_colmin_blocks(N);
    hipLaunchKernelGGL(gnm_colmin_partial_kernel, dim3(nblk), dim3(kMaxpoolThreads), 0, s, h, ldh, N, F, ws_val, ws_idx);
    GNM_CHECK_LAUNCH();
    hipLaunchKernelGGL(gnm_colmin_final_kernel, dim3((F + 63) / 64), dim3(1024), 0, s, ws_val, ws_idx, nblk, F, vmin, amin);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

extern "C" int gnm_maxpool_fwd(const float* h, int ldh, const int* nb_off, const int* nb_col, int N, int F, int max_deg,
                               int self_last, const float* eps, const float* dummy, float* out, int ldo, int* amax,
                               void* stream) {
    if (N < 0 || F <= 0 || max_deg < 0) return GNM_ERR_BAD_ARG;
    if (N == 0) return GNM_OK;
    if (!h || !nb_off || !out || ldh < F || ldo < F || (max_deg > 0 && !nb_col)) return GNM_ERR_BAD_ARG;
    // torch.max over an empty dimension raises (every node isolated and no self candidate): so does this
    if (max_deg == 0 && !self_last) return GNM_ERR_BAD_ARG;
    const long long total = (long long)N * F;
    const long long blocks = (total + kMaxpoolThreads - 1) / kMaxpoolThreads;
    if (blocks > 0x7fffffffLL) return GNM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gnm_maxpool_fwd_kernel, dim3((unsigned)blocks), dim3(kMaxpoolThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), h, ldh, nb_off, nb_col, total, F, max_deg, self_last ? 1 : 0,
                       eps, dummy, out, ldo, amax);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

extern "C" int gnm_maxpool_bwd(const float* g, int ldg, const int* amax, const int* t_off, const int* t_col, int N, int F,
                               const float* eps, const int* iso_rows, int n_iso, const int* amin, float* dh, int ldd,
                               void* stream) {
    if (N < 0 || F <= 0 || n_iso < 0) return GNM_ERR_BAD_ARG;
    if (N == 0) return GNM_OK;
    if (!g || !amax || !t_off || !dh || ldg < F || ldd < F || (n_iso > 0 && (!iso_rows || !amin))) return GNM_ERR_BAD_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long total = (long long)N * F;
    const long long blocks = (total + kMaxpoolThreads - 1) / kMaxpoolThreads;
    if (blocks > 0x7fffffffLL) return GNM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gnm_maxpool_bwd_kernel, dim3((unsigned)blocks), dim3(kMaxpoolThreads), 0, s, g, ldg, amax, t_off,
                       t_col, total, F, eps, dh, ldd);
    GNM_CHECK_LAUNCH();
    if (n_iso > 0) {
        hipLaunchKernelGGL(gnm_maxpool_bwd_dummy_kernel, dim3((F + 63) / 64), dim3(64), 0, s, g, ldg, amax, iso_rows, n_iso,
                           amin, F, dh, ldd);
        GNM_CHECK_LAUNCH();
    }
    return GNM_OK;
}

// The tiled forms (one workgroup per graph, rows in LDS): F = 32 or 64, 16-byte aligned rows, a graph's tile within
// the CU's LDS (forward 4 F n_max bytes, backward 6 F n_max); anything else returns GNM_ERR_UNSUPPORTED and the caller
// uses gnm_maxpool_fwd / gnm_maxpool_bwd, which give the same bits.
extern "C" int gnm_maxpool_fwd_tiled(const float* h, int ldh, const int* nb_off, const int* nb_col, const int* node_off, int B,
                                     int n_max, int F, int max_deg, int self_last, const float* eps, const float* dummy,
                                     float* out, int ldo, int* amax, void* stream) {
    if (B < 0 || F <= 0 || max_deg < 0) return GNM_ERR_BAD_ARG;
    if (B == 0) return GNM_OK;
    if (!h || !nb_off || !node_off || !out || ldh < F || ldo < F || (max_deg > 0 && !nb_col)) return GNM_ERR_BAD_ARG;
    if (max_deg == 0 && !self_last) return GNM_ERR_BAD_ARG;
    const size_t lds = (size_t)(n_max + 1) * F * 4 + (size_t)(n_max + 1) * 4;
    if (!maxpool_tile_ok(h, ldh, out, ldo, F, n_max, lds) || (dummy && (reinterpret_cast<uintptr_t>(dummy) & 15)) ||
        (amax && (reinterpret_cast<uintptr_t>(amax) & 15)))
        return GNM_ERR_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (F == 64) {
        GNM_ALLOW_FULL_LDS(&gnm_maxpool_tile_fwd_kernel<16>);
        hipLaunchKernelGGL(gnm_maxpool_tile_fwd_kernel<16>, dim3(B), dim3(kMaxTileThreads), lds, s, h, ldh, nb_off, nb_col,
                           node_off, max_deg, self_last ? 1 : 0, eps, dummy, out, ldo, amax);
    } else {
        GNM_ALLOW_FULL_LDS(&gnm_maxpool_tile_fwd_kernel<8>);
        hipLaunchKernelGGL(gnm_maxpool_tile_fwd_kernel<8>, dim3(B), dim3(kMaxTileThreads), lds, s, h, ldh, nb_off, nb_col,
                           node_off, max_deg, self_last ? 1 : 0, eps, dummy, out, ldo, amax);
    }
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

extern "C" int gnm_maxpool_bwd_tiled(const float* g, int ldg, const int* amax, const int* t_off, const int* t_col,
                                     const int* node_off, int B, int n_max, int F, const float* eps, const int* iso_rows,
                                     int n_iso, const int* amin, float* dh, int ldd, void* stream) {
    if (B < 0 || F <= 0 || n_iso < 0) return GNM_ERR_BAD_ARG;
    if (B == 0) return GNM_OK;
    if (!g || !amax || !t_off || !node_off || !dh || ldg < F || ldd < F || (n_iso > 0 && (!iso_rows || !amin)))
        return GNM_ERR_BAD_ARG;
    const size_t lds = (size_t)(n_max + 1) * F * 6 + (size_t)(n_max + 1) * 4;
    if (!maxpool_tile_ok(g, ldg, dh, ldd, F, n_max, lds) || (reinterpret_cast<uintptr_t>(amax) & 15)) return GNM_ERR_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (F == 64) {
        GNM_ALLOW_FULL_LDS(&gnm_maxpool_tile_bwd_kernel<16>);
        hipLaunchKernelGGL(gnm_maxpool_tile_bwd_kernel<16>, dim3(B), dim3(kMaxTileThreads), lds, s, g, ldg, amax, t_off, t_col,
                           node_off, eps, dh, ldd);
    } else {
        GNM_ALLOW_FULL_LDS(&gnm_maxpool_tile_bwd_kernel<8>);
        hipLaunchKernelGGL(gnm_maxpool_tile_bwd_kernel<8>, dim3(B), dim3(kMaxTileThreads), lds, s, g, ldg, amax, t_off, t_col,
                           node_off, eps, dh, ldd);
    }
    GNM_CHECK_LAUNCH();
    if (n_iso > 0) {
        hipLaunchKernelGGL(gnm_maxpool_bwd_dummy_kernel, dim3((F + 63) / 64), dim3(64), 0, s, g, ldg, amax, iso_rows, n_iso,
                           amin, F, dh, ldd);
        GNM_CHECK_LAUNCH();
    }
    return GNM_OK;
}
