// Train-step tail for gfx950: the two losses of the reference's train() and its Adam update, each as ONE pass
// over flat buffers.
//
//   loss = CrossEntropyLoss(c_logit, labels) + beta * BCEWithLogitsLoss(d_logit, d_labels)      main.py:16-17,34-37
//   optimizer = Adam(model.parameters(), lr)                                                  main.py:136, 39-41
//
// The arithmetic is PyTorch's (third party, not under /root/reference): mean-reduced log-softmax NLL, the
// numerically stable BCE-with-logits form max(x,0) - x y + log1p(exp(-|x|)), and torch.optim.Adam's default (non-AMSGrad, L2
// weight decay) update  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
// Both losses also emit their gradients (dC, dD), so backward starts from them without a torch loss graph.
// Reductions are fixed-order (per-block fp64 partials, one finishing block): bitwise reproducible.
#include "gnm_common.h"

static constexpr int kBceThreads = 256;
static constexpr int kBcePerThread = 8;

// pass 1: elementwise BCE-with-logits, gradient, per-block fp64 partial
__global__ void __launch_bounds__(kBceThreads) gnm_bce_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ target, long long M,
                                                              long long n_pos, float gscale,
                                                              float* __restrict__ dD,
                                                              double* __restrict__ partial) {
    __shared__ double lds[kBceThreads / kWave];
    const long long base = (long long)blockIdx.x * (kBceThreads * kBcePerThread);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < kBcePerThread; ++j) {
        const long long i = base + (long long)j * kBceThreads + threadIdx.x;
        if (i < M) {
            const float v = x[i];
            const float y = target ? target[i] : (i < n_pos ? 1.f : 0.f);
            const float e = expf(-fabsf(v));                      // exp(-|x|) in (0,1]
            // (1-y) x + softplus(-x) written as max(x,0) - x y + log1p(exp(-|x|)): no cancellation at |x| >> 1
            acc += (double)(fmaxf(v, 0.f) - v * y) + (double)log1pf(e);
            const float sig = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            if (dD) dD[i] = gscale * (sig - y);
        }
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kBceThreads / kWave; ++w) s += lds[w];
        partial[blockIdx.x] = s;
    }
}

// pass 2 (one block): finish the BCE sum, cross entropy over [B,C] with its gradient, write the three scalars
__global__ void __launch_bounds__(256) gnm_loss_finish_kernel(const float* __restrict__ c_logit, int ldc,
                                                              const long long* __restrict__ labels, int B, int C,
                                                              const double* __restrict__ partial, int nblk,
                                                              long long M, float beta, float* __restrict__ loss3,
                                                              float* __restrict__ dC, int lddc) {
    __shared__ double lds[2][256];
    const int tid = threadIdx.x;
    double bce = 0.0, ce = 0.0;
    for (int i = tid; i < nblk; i += 256) bce += partial[i];
    const float invB = B > 0 ? 1.f / (float)B : 0.f;
    for (int r = tid; r < B; r += 256) {
        const float* row = c_logit + (long long)r * ldc;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
        const float lse = mx + logf(se);
        const long long labl = labels[r];
        const bool lab_ok = labl >= 0 && labl < C;     // torch's CrossEntropyLoss asserts on the device; here an
        const int lab = lab_ok ? (int)labl : 0;        // out-of-range label reads nothing and makes the loss NaN
        ce += lab_ok ? (double)(lse - row[lab]) : (double)__builtin_nanf("");
        if (dC) {
            const float inv = 1.f / se;
            for (int c = 0; c < C; ++c)
                dC[(long long)r * lddc + c] = (expf(row[c] - mx) * inv - (c == lab ? 1.f : 0.f)) * invB;
        }
    }
    lds[0][tid] = bce;
    lds[1][tid] = ce;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            lds[0][tid] += lds[0][tid + off];
            lds[1][tid] += lds[1][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float ce_m = B > 0 ? (float)(lds[1][0] / (double)B) : 0.f;
        const float bce_m = M > 0 ? (float)(lds[0][0] / (double)M) : 0.f;
        loss3[0] = ce_m + beta * bce_m;
        loss3[1] = ce_m;
        loss3[2] = bce_m;
    }
}

extern "C" long long gnm_loss_workspace_doubles(long long M) {
    const long long per = (long long)kBceThreads * kBcePerThread;
    long long n = (M + per - 1) / per;
    return n > 0 ? n : 1;
}

extern "C" int gnm_loss_ce_bce(const float* c_logit, int ldc, const long long* labels, int B, int C,
                               const float* d_logit, const float* d_target, long long M, long long n_pos, float beta,
                               float* loss3, float* dC, int lddc, float* dD, double* workspace, void* stream) {
    if (B < 0 || C < 1 || M < 0 || !loss3 || !workspace || (B > 0 && (!c_logit || !labels)) || (M > 0 && !d_logit))
        return GNM_ERR_BAD_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nblk = M > 0 ? (int)gnm_loss_workspace_doubles(M) : 0;
    if (nblk > 0) {
        const float gscale = beta / (float)M;
        hipLaunchKernelGGL(gnm_bce_kernel, dim3(nblk), dim3(kBceThreads), 0, st, d_logit, d_target, M, n_pos, gscale,
                           dD, workspace);
        GNM_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(gnm_loss_finish_kernel, dim3(1), dim3(256), 0, st, c_logit, ldc, labels, B, C, workspace, nblk, M,
                       beta, loss3, dC, lddc);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// Gradients only, scaled by the upstream gradient *gscale_dev (a device scalar: what autograd hands to backward),
// in ONE launch: blocks [0, nblk) the BCE part, the last block the cross-entropy part.
__global__ void __launch_bounds__(kBceThreads) gnm_loss_grad_kernel(const float* __restrict__ c_logit, int ldc,
                                                                    const long long* __restrict__ labels, int B, int C,
                                                                    const float* __restrict__ x,
                                                                    const float* __restrict__ target, long long M,
                                                                    long long n_pos, float beta,
                                                                    const float* __restrict__ gscale_dev,
                                                                    float* __restrict__ dC, int lddc,
                                                                    float* __restrict__ dD, int nblk) {
    const float up = gscale_dev ? *gscale_dev : 1.f;
    if ((int)blockIdx.x < nblk) {
        const float gscale = up * (beta / (float)M);
        const long long base = (long long)blockIdx.x * (kBceThreads * kBcePerThread);
#pragma unroll
        for (int j = 0; j < kBcePerThread; ++j) {
            const long long i = base + (long long)j * kBceThreads + threadIdx.x;
            if (i < M) {
                const float v = x[i];
                const float y = target ? target[i] : (i < n_pos ? 1.f : 0.f);
                const float e = expf(-fabsf(v));
                const float sig = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
                dD[i] = gscale * (sig - y);
            }
        }
        return;
    }
    if (!dC) return;
    const float invB = B > 0 ? up / (float)B : 0.f;
    for (int r = threadIdx.x; r < B; r += kBceThreads) {
        const float* row = c_logit + (long long)r * ldc;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
        const float inv = 1.f / se;
        const int lab = (int)labels[r];
        for (int c = 0; c < C; ++c)
            dC[(long long)r * lddc + c] = (expf(row[c] - mx) * inv - (c == lab ? 1.f : 0.f)) * invB;
    }
}

extern "C" int gnm_loss_ce_bce_grad(const float* c_logit, int ldc, const long long* labels, int B, int C,
                                    const float* d_logit, const float* d_target, long long M, long long n_pos,
                                    float beta, const float* gscale_dev, float* dC, int lddc, float* dD,
                                    void* stream) {
    if (B < 0 || C < 1 || M < 0 || (B > 0 && dC && (!c_logit || !labels)) || (M > 0 && (!d_logit || !dD)))
        return GNM_ERR_BAD_ARG;
    const int nblk = M > 0 ? (int)gnm_loss_workspace_doubles(M) : 0;
    hipLaunchKernelGGL(gnm_loss_grad_kernel, dim3(nblk + 1), dim3(kBceThreads), 0, reinterpret_cast<hipStream_t>(stream),
                       c_logit, ldc, labels, B, C, d_logit, d_target, M, n_pos, beta, gscale_dev, dC, lddc, dD, nblk);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Adam over a flat parameter buffer.  hyper (device, 6 doubles -- Python floats, as torch.optim holds them): lr,
// beta1, beta2, eps, weight_decay, grad_scale;
// step (device int32): updates done so far; incremented by a second 1-thread launch on the same stream, so the
// whole update can sit inside a captured hipGraph and StepLR is a 4-byte write between replays.
__global__ void __launch_bounds__(256) gnm_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long long n,
                                                       const double* __restrict__ hyper,
                                                       const int* __restrict__ step) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double lr = hyper[0], b1 = hyper[1], b2 = hyper[2];
    const float eps = (float)hyper[3], wd = (float)hyper[4], gs = (float)hyper[5];
    const int t = *step + 1;
    const double bc1 = 1.0 - pow(b1, (double)t);
    const double bc2 = 1.0 - pow(b2, (double)t);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - b1), w2 = (float)(1.0 - b2), b2f = (float)b2;
    float pi = p[i];
    float gi = g[i] * gs;
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = m[i] + w1 * (gi - m[i]);                       // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2f + w2 * gi * gi;                     // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
}

__global__ void gnm_step_inc_kernel(int* step) { *step += 1; }

extern "C" int gnm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                             const double* hyper, int* step, void* stream) {
    if (n < 0 || !hyper || !step || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return GNM_ERR_BAD_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n > 0) {
        hipLaunchKernelGGL(gnm_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, param, grad, exp_avg,
                           exp_avg_sq, n, hyper, step);
        GNM_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(gnm_step_inc_kernel, dim3(1), dim3(1), 0, st, step);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}
