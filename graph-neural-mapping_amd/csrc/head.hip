// Graph-level head of GIN_InfoMaxReg for gfx950: the per-layer classifier Linears with dropout summed into
// c_logit (graphcnn.py:224-231) and the sigmoid of the graph summary that feeds the discriminator
// (graphcnn.py:239) -- work on [B, L*H] only, i.e. a few hundred KB.  As separate torch ops it is ~25 launches of
// 4-20 us per step (stack, baddbmm, mul, sum, sigmoid, bmm, expand, copies into the flat gradient buffer, ...):
// at 512 graphs per GPU that is ~6 % of the step.  Here: ONE launch forward, ONE backward.
#include "gnm_common.h"

#define GNM_MAX_LAYERS 16
struct WPtrs {
    const float* w[GNM_MAX_LAYERS];   // linears_prediction[l].weight, [C,H] row-major
    const float* b[GNM_MAX_LAYERS];   // linears_prediction[l].bias, [C]
};
struct GPtrs {
    float* w[GNM_MAX_LAYERS];
    float* b[GNM_MAX_LAYERS];
};

static constexpr int kHeadThreads = 256;
static constexpr int kHeadCols = 8;                               // columns of one dWp row per gradient block
static constexpr int kHeadSlices = kHeadThreads / kHeadCols;      // batch slices summed in parallel

// One workgroup per graph b:
//   csig[b, j]    = sigmoid(g_f[b, j])                                       j < L*H
//   c_logit[b, c] = sum_l mask[l,b,c] * (g_f[b, lH:(l+1)H] . Wp[l][c,:] + bp[l][c])     (mask = 1 when NULL)
__global__ void __launch_bounds__(kHeadThreads) gnm_head_fwd_kernel(const float* __restrict__ g_f, int ldg, int B,
                                                                    int L, int H, int C, const WPtrs wp,
                                                                    const float* __restrict__ masks,
                                                                    float* __restrict__ c_logit, int ldc,
                                                                    float* __restrict__ csig, int ldcs) {
    extern __shared__ float lg[];                 // [L*C]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* g = g_f + (size_t)b * ldg;
    if (csig) {
        for (int j = tid; j < L * H; j += kHeadThreads) csig[(size_t)b * ldcs + j] = 1.f / (1.f + expf(-g[j]));
    }
    const int wave = tid / kWave, lane = tid & (kWave - 1);
    for (int p = wave; p < L * C; p += kHeadThreads / kWave) {
        const int l = p / C, c = p - l * C;
        const float* w = wp.w[l] + (size_t)c * H;
        float s = 0.f;
        for (int h = lane; h < H; h += kWave) s = fmaf(g[l * H + h], w[h], s);
        s = wave_sum(s);
        if (lane == 0) lg[p] = s + wp.b[l][c];
    }
    __syncthreads();
    if (tid < C) {
        float acc = 0.f;
        for (int l = 0; l < L; ++l) {
            const float m = masks ? masks[((size_t)l * B + b) * C + tid] : 1.f;
            acc += m * lg[l * C + tid];
        }
        c_logit[(size_t)b * ldc + tid] = acc;
    }
}

// Blocks [0, B): per graph b, the gradient wrt the graph summary, laid out like g_f:
//   dph[b, lH+h] = sum_c dC[b,c] mask[l,b,c] Wp[l][c,h]  +  T[b, lH+h] * cs (1 - cs)        cs = csig[b, lH+h]
// (second term only when T = dU Wd is given: the discriminator's path through the sigmoid).
// Blocks [B, B + L*C*ceil(H/8)): the classifier parameter gradients, reduced over the batch in a fixed order:
//   dWp[l][c,h] = sum_b dC[b,c] mask[l,b,c] g_f[b, lH+h] ;  dbp[l][c] = sum_b dC[b,c] mask[l,b,c]
__global__ void __launch_bounds__(kHeadThreads) gnm_head_bwd_kernel(const float* __restrict__ dC, int lddc,
                                                                    const float* __restrict__ masks,
                                                                    const float* __restrict__ g_f, int ldg,
                                                                    const float* __restrict__ csig, int ldcs,
                                                                    const float* __restrict__ T, int ldt, int B, int L,
                                                                    int H, int C, const WPtrs wp, const GPtrs gp,
                                                                    float* __restrict__ dph, int lddph) {
    extern __shared__ float red[];                // [kHeadThreads] (+ [kHeadThreads] for the bias sums)
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x;
        for (int j = tid; j < L * H; j += kHeadThreads) {
            const int l = j / H, h = j - l * H;
            float acc = 0.f;
            for (int c = 0; c < C; ++c) {
                const float m = masks ? masks[((size_t)l * B + b) * C + c] : 1.f;
                acc = fmaf(dC[(size_t)b * lddc + c] * m, wp.w[l][(size_t)c * H + h], acc);
            }
            if (T) {
                const float cs = csig[(size_t)b * ldcs + j];
                acc = fmaf(T[(size_t)b * ldt + j], cs * (1.f - cs), acc);
            }
            dph[(size_t)b * lddph + j] = acc;
        }
        return;
    }
    // parameter gradients: one block per (layer, class, group of 8 columns); 32 batch slices per column run in
    // parallel (the sum over B is a latency chain otherwise) and are combined in slice order through LDS
    const int hgroups = (H + kHeadCols - 1) / kHeadCols;
    int q = blockIdx.x - B;
    const int l = q / (C * hgroups);
    q -= l * C * hgroups;
    const int c = q / hgroups, h0 = (q - c * hgroups) * kHeadCols;
    const int hh = tid % kHeadCols, sl = tid / kHeadCols;          // sl < kHeadSlices
    const int h = h0 + hh;
    float acc = 0.f, accb = 0.f;
    if (h < H) {
#pragma unroll 4
        for (int b = sl; b < B; b += kHeadSlices) {
            const float m = masks ? masks[((size_t)l * B + b) * C + c] : 1.f;
            const float d = dC[(size_t)b * lddc + c] * m;
            acc = fmaf(d, g_f[(size_t)b * ldg + l * H + h], acc);
            accb += d;
        }
    }
    red[tid] = acc;
    red[kHeadThreads + tid] = accb;
    __syncthreads();
    if (tid < kHeadCols && h < H) {
        float a = 0.f, ab = 0.f;
        for (int k = 0; k < kHeadSlices; ++k) {
            a += red[k * kHeadCols + tid];
            ab += red[kHeadThreads + k * kHeadCols + tid];
        }
        gp.w[l][(size_t)c * H + h] = a;
        if (h == 0) gp.b[l][c] = ab;
    }
}

static bool head_shape_ok(int B, int L, int H, int C) {
    return B >= 0 && L >= 1 && L <= GNM_MAX_LAYERS && H >= 1 && C >= 1 && C <= kHeadThreads && L * C <= 4096;
}

extern "C" int gnm_head_fwd(const float* g_f, int ldg, int B, int L, int H, int C, const float* const* wp_host,
                            const float* const* bp_host, const float* masks, float* c_logit, int ldc, float* csig,
                            int ldcs, void* stream) {
    if (!head_shape_ok(B, L, H, C)) return GNM_ERR_UNSUPPORTED;
    if (B == 0) return GNM_OK;
    if (!g_f || !wp_host || !bp_host || !c_logit) return GNM_ERR_BAD_ARG;
    WPtrs wp;
    for (int l = 0; l < GNM_MAX_LAYERS; ++l) {
        wp.w[l] = l < L ? wp_host[l] : nullptr;
        wp.b[l] = l < L ? bp_host[l] : nullptr;
    }
    hipLaunchKernelGGL(gnm_head_fwd_kernel, dim3(B), dim3(kHeadThreads), (size_t)L * C * 4,
                       reinterpret_cast<hipStream_t>(stream), g_f, ldg, B, L, H, C, wp, masks, c_logit, ldc, csig, ldcs);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

extern "C" int gnm_head_bwd(const float* dC, int lddc, const float* masks, const float* g_f, int ldg,
                            const float* csig, int ldcs, const float* T, int ldt, int B, int L, int H, int C,
                            const float* const* wp_host, float* const* dwp_host, float* const* dbp_host, float* dph,
                            int lddph, void* stream) {
    if (!head_shape_ok(B, L, H, C)) return GNM_ERR_UNSUPPORTED;
    if (B == 0) return GNM_OK;
    if (!dC || !g_f || !wp_host || !dwp_host || !dbp_host || !dph || (T && !csig)) return GNM_ERR_BAD_ARG;
    WPtrs wp;
    GPtrs gp;
    for (int l = 0; l < GNM_MAX_LAYERS; ++l) {
        wp.w[l] = l < L ? wp_host[l] : nullptr;
        wp.b[l] = nullptr;
        gp.w[l] = l < L ? dwp_host[l] : nullptr;
        gp.b[l] = l < L ? dbp_host[l] : nullptr;
    }
    const int hgroups = (H + kHeadCols - 1) / kHeadCols;
    hipLaunchKernelGGL(gnm_head_bwd_kernel, dim3(B + L * C * hgroups), dim3(kHeadThreads), (size_t)2 * kHeadThreads * 4,
                       reinterpret_cast<hipStream_t>(stream), dC, lddc, masks, g_f, ldg, csig, ldcs, T, ldt, B, L, H, C,
                       wp, gp, dph, lddph);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

