// Deep-Graph-Infomax scorer of the GIN_InfoMaxReg tail (K10/K11 of SURVEY.md section 2.2):
// /root/reference models/discriminator.py:19-38 called from models/graphcnn.py:233-246.
//
// The reference expands every graph summary c[g] to all of its nodes and evaluates
// nn.Bilinear(LH, LH, 1) on [N, LH] x [N, LH] (2*N*2*LH^2 FLOP).  Algebraically
//     sc_1[v] = n_f[v]      . U[g(v)] + b,     U = sigmoid(g_f) W^T   ([B, LH], tiny GEMM)
//     sc_2[v] = n_f[idx[v]] . U[g(v)] + b,     idx[v] = perm[g(v)]    (a ROW index < B:
//                                                                       graphcnn.py:198-201,242)
// so the N-sized work is one HBM-bound row-dot over the L hidden layers; sc_2 is constant
// inside a graph.  n_f (torch.cat(hidden_rep, 1), graphcnn.py:233) is never materialised:
// the kernels take the L per-layer [N, H] buffers.
#include "gnm_common.h"

#define GNM_MAX_LAYERS 16
// Layer l of n_f is either given as the activation itself (sc[l] == null: p[l] = h_l) or, so that the activation
// never has to be written to HBM, as the pre-BatchNorm output Z_l of the layer's last Linear with the folded
// BatchNorm vectors: h_l = relu(Z_l * sc[l] + sh[l]) is then re-formed on the fly (graphcnn.py:163-166).
struct HPtrs {
    const float* p[GNM_MAX_LAYERS];
    const float* sc[GNM_MAX_LAYERS];
    const float* sh[GNM_MAX_LAYERS];
};
__device__ __forceinline__ float gnm_sigmoid(float v) {      // the form of csrc/tail.hip's BCE gradient
    const float e = expf(-fabsf(v));
    return v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}
__device__ __forceinline__ float4 gnm_bnrelu4(float4 x, float4 s, float4 h) {
    return make_float4(gnm_relu(x.x * s.x + h.x), gnm_relu(x.y * s.y + h.y), gnm_relu(x.z * s.z + h.z),
                       gnm_relu(x.w * s.w + h.w));
}

// d_logit[v] = sc_1[v], d_logit[N + v] = sc_2[g(v)]; one workgroup per graph.
// A row of layer l is covered by H/4 lanes with 16-B loads (G = 64/(H/4) rows per
// wave-instruction when H/4 divides 64, e.g. 4 rows at H = 64); the lane group reduces its
// dot product with DPP/shuffle steps inside the group.  Generic widths use one wave per row.
//
// UNIT (round 3): the launch also leaves the backward's per-graph reductions for the reference's own loss,
// BCEWithLogits against ones (true pairs) / zeros (shuffled pairs) (main.py:32-37), up to the loss's scalar factor k:
//     d D[v] = k (sigmoid(sc_1[v]) - 1),   d D[N + v] = k sigmoid(sc_2[g(v)])
//     unit[g, l*H + c] = sum_{v in g} (sigmoid(sc_1[v]) - 1) h_l[v, c] + n_g sigmoid(sc_2[g]) h_l[perm_rows[g], c]   (= dU / k)
//     unit[g, L*H]     = n_g sigmoid(sc_2[g])                                                                       (= s2sum / k)
//     unit[g, L*H + 1] = unit[g, L*H] + sum_{v in g} (sigmoid(sc_1[v]) - 1)                                         (= dsum / k)
//     inv_perm[perm_rows[g]] = g
// from the rows this kernel holds in registers anyway -- gnm_disc_du_kernel's second pass over the five hidden layers
// (539 MB at B = 1024) is then not needed: the backward scales `unit` by k (gnm_disc_unit_scale).  It cannot be folded
// into the aggregation-backward epilogues instead: those CONSUME the readout gradient
// dpool = f(dU W), so all of dU must exist before the first of them runs.
template <int LPR4, bool UNIT = false>   // lanes per row (H/4), a power of two <= 64; 0 = generic
// (the UNIT form holds ~150 registers: it is always launched with 256 threads)
__global__ void __launch_bounds__(UNIT ? 256 : 1024) gnm_disc_score_kernel(const HPtrs hp, int ldh, int L, int H,
                                                             const float* __restrict__ U, int ldu,
                                                             const int32_t* __restrict__ perm_rows,
                                                             const float* __restrict__ bias,
                                                             const int32_t* __restrict__ node_off, int N,
                                                             float* __restrict__ d_logit,
                                                             float* __restrict__ unit, int ldunit,
                                                             int32_t* __restrict__ inv_perm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Us = reinterpret_cast<float*>(smem);        // [L*H]
    float* Ss = Us + L * H;                            // [L*H] BatchNorm scale / shift of the layers given as Z
    float* Sh = Ss + L * H;
    float* sc2s = Sh + L * H;                          // [1]
    const int g = blockIdx.x;
    const int row0 = node_off[g];
    const int n = node_off[g + 1] - row0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;      // 256 threads, or 1024 for batches of few graphs
    unsigned tmask = 0;                                // layers given as Z (workgroup-uniform)
    for (int l = 0; l < L; ++l) tmask |= hp.sc[l] ? 1u << l : 0u;
    // The first row group of every wave is requested BEFORE the workgroup's prologue (U / BatchNorm vectors into LDS,
    // the shuffled-branch score by wave 0: a dependent chain of a few microseconds in a workgroup that lives ~20):
    // those loads depend on nothing the prologue produces (round 3).
    constexpr int MLE = 5;
    float4 xe[MLE];
    if constexpr (LPR4 > 0) {
        if (n > 0) {
            const int sub_e = lane & (LPR4 - 1), slot_e = lane / LPR4;
            const size_t vo = (size_t)(row0 + min(wave * (64 / LPR4) + slot_e, n - 1)) * ldh + 4 * sub_e;
#pragma unroll
            for (int l = 0; l < MLE; ++l) xe[l] = *reinterpret_cast<const float4*>(hp.p[min(l, L - 1)] + vo);
        }
    }
    for (int e = tid; e < L * H; e += nthreads) {
        Us[e] = U[(size_t)g * ldu + e];
        const int l = e / H, c = e - l * H;
        Ss[e] = hp.sc[l] ? hp.sc[l][c] : 1.f;
        Sh[e] = hp.sc[l] ? hp.sh[l][c] : 0.f;
    }
    __syncthreads();
    const float bv = bias ? bias[0] : 0.f;
    if (wave == 0) {
        const int pr = perm_rows[g];
        float a = 0.f;
        for (int l = 0; l < L; ++l)
            for (int c = lane; c < H; c += 64) {
                float x = hp.p[l][(size_t)pr * ldh + c];
                if (tmask >> l & 1) x = gnm_relu(x * Ss[l * H + c] + Sh[l * H + c]);
                a += x * Us[l * H + c];
            }
        a = wave_sum(a);
        if (lane == 0) sc2s[0] = a + bv;
    }
    __syncthreads();
    const float sc2 = sc2s[0];
    if (n <= 0) {
        if constexpr (UNIT) {          // an empty graph: empty sums
            for (int e = tid; e < L * H + 2; e += nthreads) unit[(size_t)g * ldunit + e] = 0.f;
            if (tid == 0) inv_perm[perm_rows[g]] = g;
        }
        return;
    }
    if constexpr (LPR4 > 0) {
        constexpr int G = 64 / LPR4;                   // rows per wave-instruction
        const int sub = lane & (LPR4 - 1), slot = lane / LPR4;
        constexpr int ML = 5;                          // layers handled with all their loads in flight at once
        float4 uu[ML];
#pragma unroll
        for (int l = 0; l < ML; ++l)
            uu[l] = l < L ? *reinterpret_cast<const float4*>(Us + l * H + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        // Two row groups in flight, ping-pong: the loads of the next group are issued BEFORE the stores of the current
        // one, every access unconditional (a layer past L re-reads layer L - 1 against a zero U; a lane that has no
        // result to write stores to an offset the buffer descriptor clips), so that the wait for a group's loads is a
        // counted vmcnt and never a drain that includes the previous group's stores (gfx950 retires vector-memory
        // operations in issue order).  The loop it replaces -- load, reduce, guarded stores, repeat -- paid an HBM
        // write round trip per 4 rows.
        const float* lp[ML];
#pragma unroll
        for (int l = 0; l < ML; ++l) lp[l] = hp.p[min(l, L - 1)];
        const __amdgpu_buffer_rsrc_t rd =
            __builtin_amdgcn_make_buffer_rsrc(d_logit, 0, (int)((unsigned)(2 * (size_t)N * 4)), 0x00020000);
        const int stride = nwaves * G;
#define GNM_DS_LOAD(xx, r_)                                                                                      \
        {                                                                                                        \
            const size_t vo = (size_t)(row0 + min((r_), n - 1)) * ldh + 4 * sub;                                 \
            _Pragma("unroll") for (int l = 0; l < ML; ++l) xx[l] = *reinterpret_cast<const float4*>(lp[l] + vo); \
        }
#define GNM_DS_FINISH(xx, r_)                                                                                    \
        {                                                                                                        \
            const int v = row0 + min((r_), n - 1);                                                               \
            float a = 0.f;                                                                                       \
            _Pragma("unroll") for (int l = 0; l < ML; ++l) {                                                     \
                float4 x = xx[l];                                                                                \
                if (tmask >> l & 1)                                                                              \
                    x = gnm_bnrelu4(x, *reinterpret_cast<const float4*>(Ss + min(l, L - 1) * H + 4 * sub),       \
                                    *reinterpret_cast<const float4*>(Sh + min(l, L - 1) * H + 4 * sub));         \
                xx[l] = x;                                                                                       \
                const float dl = x.x * uu[l].x + x.y * uu[l].y + x.z * uu[l].z + x.w * uu[l].w;                  \
                a += l < L ? dl : 0.f;      /* a padding slot re-reads layer L - 1: keep a non-finite z out */   \
            }                                                                                                    \
            for (int l = ML; l < L; ++l) {                                                                       \
                float4 x = *reinterpret_cast<const float4*>(hp.p[l] + (size_t)v * ldh + 4 * sub);                \
                if (tmask >> l & 1)                                                                              \
                    x = gnm_bnrelu4(x, *reinterpret_cast<const float4*>(Ss + l * H + 4 * sub),                   \
                                    *reinterpret_cast<const float4*>(Sh + l * H + 4 * sub));                     \
                const float4 u = *reinterpret_cast<const float4*>(Us + l * H + 4 * sub);                         \
                a += x.x * u.x + x.y * u.y + x.z * u.z + x.w * u.w;                                              \
            }                                                                                                    \
            _Pragma("unroll") for (int off = LPR4 >> 1; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);         \
            if constexpr (UNIT) {         /* every lane of the row's group holds the whole dot product */        \
                const float wv = (r_) < n ? gnm_sigmoid(a + bv) - 1.f : 0.f;                                     \
                s1u += sub == 0 ? wv : 0.f;                                                                      \
                _Pragma("unroll") for (int l = 0; l < ML; ++l) {                                                 \
                    du[l].x += wv * xx[l].x; du[l].y += wv * xx[l].y;                                            \
                    du[l].z += wv * xx[l].z; du[l].w += wv * xx[l].w;                                            \
                }                                                                                                \
            }                                                                                                    \
            const bool wr = sub == 0 && (r_) < n;                                                                \
            const unsigned o1 = wr ? (unsigned)v * 4u : 0xFFFFFFF0u;                                             \
            const unsigned o2 = wr ? (unsigned)(N + v) * 4u : 0xFFFFFFF0u;                                       \
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a + bv), rd, o1, 0, 0);                        \
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc2), rd, o2, 0, 0);                           \
        }
        float4 xa[ML], xb[ML];
        float4 du[ML];
        float s1u = 0.f;
#pragma unroll
        for (int l = 0; l < ML; ++l) du[l] = make_float4(0.f, 0.f, 0.f, 0.f);
        int r = wave * G + slot;
        static_assert(ML == MLE, "the early request covers the same layers");
#pragma unroll
        for (int l = 0; l < ML; ++l) xa[l] = xe[l];             // (requested at kernel entry)
        for (; r - slot < n; r += 2 * stride) {         // wave-uniform trip count (r - slot is the wave's first row)
            GNM_DS_LOAD(xb, r + stride)
            GNM_DS_FINISH(xa, r)
            GNM_DS_LOAD(xa, r + 2 * stride)
            GNM_DS_FINISH(xb, r + stride)
        }
#undef GNM_DS_LOAD
#undef GNM_DS_FINISH
        if constexpr (UNIT) {
            // fixed-order reduction: the row slots of a wave (lanes with equal `sub`), then the waves through LDS
            float4* red = reinterpret_cast<float4*>(sc2s + 4);            // [nwaves][ML][LPR4]
            float* s1red = reinterpret_cast<float*>(red + nwaves * ML * LPR4);
#pragma unroll
            for (int l = 0; l < ML; ++l) {
#pragma unroll
                for (int off = LPR4; off < 64; off <<= 1) {
                    du[l].x += __shfl_xor(du[l].x, off, 64); du[l].y += __shfl_xor(du[l].y, off, 64);
                    du[l].z += __shfl_xor(du[l].z, off, 64); du[l].w += __shfl_xor(du[l].w, off, 64);
                }
                if (lane < LPR4) red[(wave * ML + l) * LPR4 + lane] = du[l];
            }
            s1u = wave_sum(s1u);
            if (lane == 0) s1red[wave] = s1u;
            __syncthreads();
            const float s2u = (float)n * gnm_sigmoid(sc2);
            const int pr = perm_rows[g];
            for (int e = tid; e < L * LPR4; e += nthreads) {
                const int l = e / LPR4, c4 = e - l * LPR4;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int w = 0; w < nwaves; ++w) {
                    const float4 q = red[(w * ML + l) * LPR4 + c4];
                    t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                }
                float4 x = *reinterpret_cast<const float4*>(hp.p[l] + (size_t)pr * ldh + 4 * c4);
                if (tmask >> l & 1)
                    x = gnm_bnrelu4(x, *reinterpret_cast<const float4*>(Ss + l * H + 4 * c4),
                                    *reinterpret_cast<const float4*>(Sh + l * H + 4 * c4));
                t.x += s2u * x.x; t.y += s2u * x.y; t.z += s2u * x.z; t.w += s2u * x.w;
                *reinterpret_cast<float4*>(unit + (size_t)g * ldunit + (size_t)l * H + 4 * c4) = t;
            }
            if (tid == 0) {
                float s1 = 0.f;
                for (int w = 0; w < nwaves; ++w) s1 += s1red[w];
                unit[(size_t)g * ldunit + (size_t)L * H] = s2u;
                unit[(size_t)g * ldunit + (size_t)L * H + 1] = s2u + s1;
                inv_perm[pr] = g;
            }
        }
    } else {
        for (int r = wave; r < n; r += nwaves) {
            const int v = row0 + r;
            float a = 0.f;
            for (int l = 0; l < L; ++l)
                for (int c = lane; c < H; c += 64) {
                    float x = hp.p[l][(size_t)v * ldh + c];
                    if (tmask >> l & 1) x = gnm_relu(x * Ss[l * H + c] + Sh[l * H + c]);
                    a += x * Us[l * H + c];
                }
            a = wave_sum(a);
            if (lane == 0) {
                d_logit[v] = a + bv;
                d_logit[(size_t)N + v] = sc2;
            }
        }
    }
}

static void fill_hptrs(HPtrs& hp, const float* const* hptrs, const float* const* scale_ptrs,
                       const float* const* shift_ptrs, int L) {
    for (int l = 0; l < GNM_MAX_LAYERS; ++l) {
        hp.p[l] = l < L ? hptrs[l] : nullptr;
        const bool z = l < L && scale_ptrs && shift_ptrs && scale_ptrs[l] && shift_ptrs[l];
        hp.sc[l] = z ? scale_ptrs[l] : nullptr;
        hp.sh[l] = z ? shift_ptrs[l] : nullptr;
    }
}

static int disc_score_launch(const float* const* hptrs, const float* const* scale_ptrs, const float* const* shift_ptrs,
                             int ldh, int L, int H, const float* U, int ldu, const int32_t* perm_rows,
                             const float* bias, const int32_t* node_off, int N, int B, float* d_logit, float* unit,
                             int ldunit, int32_t* inv_perm, void* stream) {
    if (B <= 0) return GNM_OK;
    if (L <= 0 || L > GNM_MAX_LAYERS || H <= 0) return GNM_ERR_BAD_ARG;
    HPtrs hp;
    fill_hptrs(hp, hptrs, scale_ptrs, shift_ptrs, L);
    const int threads = (unit || B >= 1024) ? 256 : 1024;
    size_t lds = (size_t)(3 * L * H + 4) * 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool vec = ((ldh & 3) == 0) && ((H & 3) == 0);
    const int lpr4 = vec ? H / 4 : 0;
    if (unit) {
        // the by-products ride on the register-resident row groups of the vector forms (at most 5 layers in flight)
        if (!(lpr4 == 8 || lpr4 == 16 || lpr4 == 32) || L > 5 || (ldunit & 3) || ldunit < L * H + 2 || !inv_perm ||
            (reinterpret_cast<uintptr_t>(unit) & 15))
            return GNM_ERR_UNSUPPORTED;
        lds += 16 + (size_t)(threads / 64) * 5 * lpr4 * 16 + (size_t)(threads / 64) * 4;
    }
#define GNM_DISC_CASE(V, UN)                                                                                          \
    hipLaunchKernelGGL((gnm_disc_score_kernel<V, UN>), dim3(B), dim3(threads), lds, st, hp, ldh, L, H, U, ldu,          \
                       perm_rows, bias, node_off, N, d_logit, unit, ldunit, inv_perm)
    if (unit) {
        if (lpr4 == 8) GNM_DISC_CASE(8, true);
        else if (lpr4 == 16) GNM_DISC_CASE(16, true);
        else GNM_DISC_CASE(32, true);
    } else if (lpr4 == 8) GNM_DISC_CASE(8, false);
    else if (lpr4 == 16) GNM_DISC_CASE(16, false);
    else if (lpr4 == 32) GNM_DISC_CASE(32, false);
    else GNM_DISC_CASE(0, false);
#undef GNM_DISC_CASE
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

extern "C" int gnm_disc_score_fwd(const float* const* hptrs, const float* const* scale_ptrs,
                                  const float* const* shift_ptrs, int ldh, int L, int H, const float* U, int ldu,
                                  const int32_t* perm_rows, const float* bias, const int32_t* node_off, int N, int B,
                                  float* d_logit, void* stream) {
    return disc_score_launch(hptrs, scale_ptrs, shift_ptrs, ldh, L, H, U, ldu, perm_rows, bias, node_off, N, B,
                             d_logit, nullptr, 0, nullptr, stream);
}

// gnm_disc_score_fwd that also leaves the backward's reductions for the reference's BCE-with-logits loss up to its
// scalar factor (see the kernel): unit [B, ldunit >= L*H + 2], inv_perm [B].  GNM_ERR_UNSUPPORTED outside the vector
// forms (H / 4 in {8, 16, 32}, L <= 5): call gnm_disc_score_fwd and, in the backward, gnm_disc_score_bwd.
extern "C" int gnm_disc_score_fwd_unit(const float* const* hptrs, const float* const* scale_ptrs,
                                       const float* const* shift_ptrs, int ldh, int L, int H, const float* U, int ldu,
                                       const int32_t* perm_rows, const float* bias, const int32_t* node_off, int N,
                                       int B, float* d_logit, float* unit, int ldunit, int32_t* inv_perm,
                                       void* stream) {
    if (!unit) return GNM_ERR_BAD_ARG;
    return disc_score_launch(hptrs, scale_ptrs, shift_ptrs, ldh, L, H, U, ldu, perm_rows, bias, node_off, N, B,
                             d_logit, unit, ldunit, inv_perm, stream);
}

// dU = k unit[:, :LH], s2sum = k unit[:, LH], dsum = k unit[:, LH + 1] with k = *k_dev x kscale: the upstream gradient (a
// device scalar) times the loss's host-side factor -- multiplied here, not by an element-wise launch of the caller.  What gnm_disc_score_bwd would have produced from dD = k (sigmoid(d_logit) - target).
__global__ void __launch_bounds__(256) gnm_disc_unit_scale_kernel(const float* __restrict__ unit, int ldunit, int LH,
                                                                  const float* __restrict__ k, float kscale,
                                                                  float* __restrict__ dU, int ldu,
                                                                  float* __restrict__ s2sum, float* __restrict__ dsum,
                                                                  float* __restrict__ dbias, int B) {
    __shared__ float red[256];
    const int g = blockIdx.x;
    const float kv = *k * kscale;                 // (one fp32 product, as the loss-gradient kernel forms its factor)
    const float* row = unit + (size_t)g * ldunit;
    for (int e = threadIdx.x; e < LH; e += blockDim.x) dU[(size_t)g * ldu + e] = kv * row[e];
    if (threadIdx.x == 0) {
        s2sum[g] = kv * row[LH];
        if (dsum) dsum[g] = kv * row[LH + 1];
    }
    // the Bilinear bias gradient = sum over the graphs of dsum (discriminator.py:19: one scalar): workgroup 0 adds the B
    // values itself, in a fixed order (strided partials, then a tree) -- a torch.sum launch of the caller before
    if (dbias && g == 0) {
        float s = 0.f;
        for (int q = threadIdx.x; q < B; q += 256) s += kv * unit[(size_t)q * ldunit + LH + 1];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) *dbias = red[0];
    }
}

extern "C" int gnm_disc_unit_scale(const float* unit, int ldunit, int LH, const float* k, float kscale, int B, float* dU,
                                   int ldu, float* s2sum, float* dsum, float* dbias, void* stream) {
    if (B <= 0) return GNM_OK;
    if (!unit || !k || !dU || !s2sum || LH <= 0 || ldunit < LH + 2) return GNM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gnm_disc_unit_scale_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), unit,
                       ldunit, LH, k, kscale, dU, ldu, s2sum, dsum, dbias, B);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// Backward wrt U (and the per-graph sum of the negative-branch gradient):
//   s2sum[g]  = sum_{v in g} dD[N + v]
//   dU[g, l*H + c] = sum_{v in g} dD[v] * h_l[v, c] + s2sum[g] * h_l[perm_rows[g], c]
// (the gradient wrt n_f itself is folded into gnm_bn_relu_bwd_stats).  Optional by-products for the caller:
//   dsum[g] = sum_{v in g} (dD[v] + dD[N + v])   (their sum over g is the Bilinear bias gradient)
//   inv_perm[perm_rows[g]] = g                     (who uses graph g's first-rows as negatives)
__global__ void __launch_bounds__(1024) gnm_disc_du_kernel(const HPtrs hp, int ldh, int L, int H,
                                                          const float* __restrict__ dD,
                                                          const int32_t* __restrict__ perm_rows,
                                                          const int32_t* __restrict__ node_off, int N,
                                                          float* __restrict__ dU, int ldu,
                                                          float* __restrict__ s2sum, float* __restrict__ dsum,
                                                          int32_t* __restrict__ inv_perm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* red = reinterpret_cast<float4*>(smem);     // [RP][H4]
    __shared__ float wsum[32];                          // [2][up to 16 waves]
    const int g = blockIdx.x;
    const int row0 = node_off[g];
    const int n = node_off[g + 1] - row0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    float s = 0.f, s1 = 0.f;
    for (int r = tid; r < n; r += nthreads) {
        s += dD[(size_t)N + row0 + r];
        s1 += dD[(size_t)row0 + r];
    }
    s = wave_sum(s);
    s1 = wave_sum(s1);
    if (lane == 0) {
        wsum[wave] = s;
        wsum[16 + wave] = s1;
    }
    __syncthreads();
    float s2 = 0.f, s1t = 0.f;
    for (int w = 0; w < nwaves; ++w) {                  // fixed order
        s2 += wsum[w];
        s1t += wsum[16 + w];
    }
    if (tid == 0) {
        s2sum[g] = s2;
        if (dsum) dsum[g] = s2 + s1t;
        if (inv_perm) inv_perm[perm_rows[g]] = g;
    }
    const int H4 = H >> 2;
    const int RP = nthreads / H4;
    const int rg = tid / H4, c4 = tid - rg * H4;
    const int pr = perm_rows[g];
    for (int l = 0; l < L; ++l) {
        const float* hl = hp.p[l];
        const bool asz = hp.sc[l] != nullptr;               // this layer is given as Z + folded BatchNorm
        float4 lsc = make_float4(1.f, 1.f, 1.f, 1.f), lsh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (asz && c4 < H4) {
            lsc = *reinterpret_cast<const float4*>(hp.sc[l] + 4 * c4);
            lsh = *reinterpret_cast<const float4*>(hp.sh[l] + 4 * c4);
        }
        if (rg < RP) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int r = rg;
            for (; r + 3 * RP < n; r += 4 * RP) {          // 4 rows per thread with their loads in flight together
                float w[4];
                float4 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int v = row0 + r + u * RP;
                    w[u] = dD[v];
                    x[u] = *reinterpret_cast<const float4*>(hl + (size_t)v * ldh + 4 * c4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (asz) x[u] = gnm_bnrelu4(x[u], lsc, lsh);
                    acc.x += w[u] * x[u].x; acc.y += w[u] * x[u].y; acc.z += w[u] * x[u].z; acc.w += w[u] * x[u].w;
                }
            }
            for (; r < n; r += RP) {
                const int v = row0 + r;
                const float w = dD[v];
                float4 x = *reinterpret_cast<const float4*>(hl + (size_t)v * ldh + 4 * c4);
                if (asz) x = gnm_bnrelu4(x, lsc, lsh);
                acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
            }
            red[rg * H4 + c4] = acc;
        }
        __syncthreads();
        if (tid < H4) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < RP; ++q) {
                const float4 x = red[q * H4 + tid];
                t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
            }
            float4 x = *reinterpret_cast<const float4*>(hl + (size_t)pr * ldh + 4 * tid);
            if (asz) x = gnm_bnrelu4(x, *reinterpret_cast<const float4*>(hp.sc[l] + 4 * tid),
                                     *reinterpret_cast<const float4*>(hp.sh[l] + 4 * tid));
            t.x += s2 * x.x; t.y += s2 * x.y; t.z += s2 * x.z; t.w += s2 * x.w;
            *reinterpret_cast<float4*>(dU + (size_t)g * ldu + (size_t)l * H + 4 * tid) = t;
        }
        __syncthreads();
    }
}

extern "C" int gnm_disc_score_bwd(const float* const* hptrs, const float* const* scale_ptrs,
                                  const float* const* shift_ptrs, int ldh, int L, int H, const float* dD,
                                  const int32_t* perm_rows, const int32_t* node_off, int N, int B, float* dU, int ldu,
                                  float* s2sum, float* dsum, int32_t* inv_perm, void* stream) {
    if (B <= 0) return GNM_OK;
    if (L <= 0 || L > GNM_MAX_LAYERS || H <= 0 || (H & 3) || H > 1024 || (ldh & 3) || (ldu & 3))
        return GNM_ERR_BAD_ARG;
    HPtrs hp;
    fill_hptrs(hp, hptrs, scale_ptrs, shift_ptrs, L);
    const int threads = B >= 1024 ? 256 : 1024;       // one workgroup per graph: few graphs -> big workgroups
    const int H4 = H >> 2, RP = threads / H4;
    hipLaunchKernelGGL(gnm_disc_du_kernel, dim3(B), dim3(threads), (size_t)RP * H4 * 16,
                       reinterpret_cast<hipStream_t>(stream), hp, ldh, L, H, dD, perm_rows, node_off, N, dU, ldu,
                       s2sum, dsum, inv_perm);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}
