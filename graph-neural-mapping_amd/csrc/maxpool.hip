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
static constexpr int kColminRows = 1024;          // rows per workgroup of the column-minimum pass

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
            for (int r = r0 + phase; r < r1; r += 4) {
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

__global__ void gnm_colmin_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx, int nblk, int F,
                                        float* __restrict__ vmin, int* __restrict__ amin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= F) return;
    float bv = pval[c];
    int bi = pidx[c];
    for (int b = 1; b < nblk; ++b) {
        const float v = pval[(size_t)b * F + c];
        const int i = pidx[(size_t)b * F + c];
        if (i >= 0 && (bi < 0 || colmin_better(v, i, bv, bi))) { bv = v; bi = i; }
    }
    vmin[c] = bv;
    amin[c] = bi;
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
    if (eps) acc += (1.0f + eps[0]) * g[(size_t)j * ldg + c];
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

extern "C" int gnm_maxpool_colmin_blocks(int N) { return N > 0 ? (N + kColminRows - 1) / kColminRows : 1; }

extern "C" int gnm_maxpool_colmin(const float* h, int ldh, int N, int F, float* ws_val, int* ws_idx, float* vmin,
                                  int* amin, void* stream) {
    if (N <= 0 || F <= 0 || !h || !ws_val || !ws_idx || !vmin || !amin || ldh < F) return GNM_ERR_BAD_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nblk = gnm_maxpool_colmin_blocks(N);
    hipLaunchKernelGGL(gnm_colmin_partial_kernel, dim3(nblk), dim3(kMaxpoolThreads), 0, s, h, ldh, N, F, ws_val, ws_idx);
    GNM_CHECK_LAUNCH();
    hipLaunchKernelGGL(gnm_colmin_final_kernel, dim3((F + 63) / 64), dim3(64), 0, s, ws_val, ws_idx, nblk, F, vmin, amin);
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
