// BatchNorm1d + ReLU + graph readout passes of the GIN layer (K5, K6, K7 of SURVEY.md
// section 2.2): /root/reference models/mlp.py:38,48 and models/graphcnn.py:51,163,166,
// 187,190 (BatchNorm/ReLU) and :228-229 (readout spmm(graph_pool, h)), forward and
// backward.  All of these are HBM-bound streaming passes over [N, H] fp32 arrays with
// 16-B per-lane accesses; per-column reductions are done in registers/LDS in a fixed
// order and accumulated in fp64 so that results are reproducible run to run.
//
// Training-mode statistics arrive as per-block (sum, sum of squares) partials from the
// Linear kernel's epilogue (linear.hip); gnm_bn_finalize turns them into the
// per-column scale/shift used by the fused consumers and updates the running
// statistics exactly as torch.nn.BatchNorm1d does (biased variance to normalise,
// unbiased for running_var, momentum 0.1, num_batches_tracked += 1).
#include "gnm_common.h"

// ---------------------------------------------------------------------------------
// column-partial reduction shared by the two finalize kernels.
// partial: [nblk][2][H] doubles.  A workgroup owns a slice of CW = 4 columns (both the "sum" and the "second sum"
// halves), so the grid is H/4 workgroups of 128 row groups each: with ~768 partial rows a thread has six values to
// fetch and requests them all at once.  (One workgroup reading all ~0.8 MB of partials took ~16 us; H/16 workgroups
// whose threads walked 24 rows in dependent batches of four 6.7 us -- twenty such launches per step.)  Fixed order.
// ---------------------------------------------------------------------------------
static constexpr int kFinCols = 4;

// workgroup size of the one-workgroup-per-graph kernels: 256 threads when there are plenty of graphs, 1024 when a
// batch of few large graphs (e.g. 256 x 1000 nodes) would otherwise put 4 waves on each CU
static inline int per_graph_threads(int B) { return B >= 1024 ? 256 : 1024; }

__device__ __forceinline__ void reduce_partials_slice(const double* partial, int nblk, int H, int c0, double* lds,
                                                      double* tot /*[2*kFinCols]*/) {
    constexpr int NC = 2 * kFinCols;              // columns handled here: 4 of each half
    const int tid = threadIdx.x;                  // 1024 threads = 128 row groups x 8 columns
    const int g = tid / NC, cc = tid - g * NC;
    const int which = cc / kFinCols, c = c0 + (cc - which * kFinCols);
    constexpr int groups = 1024 / NC;
    constexpr int U = 8;
    double s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = 0.0;
    if (c < H) {
        const double* base = partial + (size_t)which * H + c;
        for (int b = g; b < nblk; b += U * groups) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int bb = b + u * groups;
                v[u] = bb < nblk ? base[(size_t)bb * 2 * H] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] += v[u];
        }
    }
    lds[g * NC + cc] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    // 128 row groups per column: four threads take 32 each, then one adds the four
    constexpr int Q = 4, GQ = groups / Q;
    double t = 0.0;
    if (tid < Q * NC) {
        const int qq = tid / NC, col = tid - qq * NC;
        for (int gg = qq * GQ; gg < (qq + 1) * GQ; ++gg) t += lds[gg * NC + col];
    }
    __syncthreads();
    if (tid < Q * NC) lds[tid] = t;
    __syncthreads();
    if (tid < NC) tot[tid] = (lds[tid] + lds[NC + tid]) + (lds[2 * NC + tid] + lds[3 * NC + tid]);
    __syncthreads();
}

__global__ void __launch_bounds__(1024) gnm_bn_finalize_kernel(
    const double* __restrict__ partial, int nblk, int H, long long nrows, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, long long* num_batches_tracked,
    float momentum, float eps, int training, int update_running, float* __restrict__ scale,
    float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double lds[1024];
    __shared__ double tot[2 * kFinCols];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * kFinCols;
    if (training) reduce_partials_slice(partial, nblk, H, c0, lds, tot);
    const int c = c0 + tid;
    if (tid < kFinCols && c < H) {
        double mean, var;
        if (training) {
            mean = tot[tid] / (double)nrows;
            var = tot[kFinCols + tid] / (double)nrows - mean * mean;
            if (var < 0.0) var = 0.0;
            if (update_running) {
                const double unbiased = nrows > 1 ? var * ((double)nrows / (double)(nrows - 1)) : var;
                running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
                running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
            }
        } else {
            mean = (double)running_mean[c];
            var = (double)running_var[c];
        }
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rstd;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        mean_out[c] = (float)mean;
        rstd_out[c] = rstd;
    }
    if (blockIdx.x == 0 && tid == 0 && training && update_running && num_batches_tracked) *num_batches_tracked += 1;
}

extern "C" int gnm_bn_finalize(const double* stats_partial, int nblk, int H, long long nrows, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               long long* num_batches_tracked, float momentum, float eps, int training,
                               int update_running, float* scale, float* shift, float* mean_out, float* rstd_out,
                               void* stream) {
    if (H <= 0 || H > 128) return GNM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gnm_bn_finalize_kernel, dim3((H + kFinCols - 1) / kFinCols), dim3(1024), 0,
                       reinterpret_cast<hipStream_t>(stream), stats_partial, nblk, H, nrows, gamma, beta, running_mean, running_var, num_batches_tracked,
                       momentum, eps, training, update_running, scale, shift, mean_out, rstd_out);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------
// y = relu(z*scale + shift) written to Hout, fused with the graph readout
// pooled[b] = sum_{v in graph b} y[v]  (x 1/n_b for "average", graphcnn.py:122-127).
// One workgroup per graph; a thread owns one 16-B column chunk of every RP-th row.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) gnm_bn_relu_readout_kernel(
    const float* __restrict__ Z, int ldz, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ Hout, int ldh, const int32_t* __restrict__ node_off, int H, int relu,
    float* __restrict__ pooled, int ldp, int average) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* red = reinterpret_cast<float4*>(smem);   // [RP][H4]
    const int b = blockIdx.x;
    const int row0 = node_off[b];
    const int n = node_off[b + 1] - row0;
    const int H4 = H >> 2;
    const int RP = (int)blockDim.x / H4;      // 256 threads, or 1024 when few graphs share the GPU (see the launcher)
    const int tid = threadIdx.x;
    const int rg = tid / H4, c4 = tid - rg * H4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rg < RP) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * c4);
        const float4 sh = *reinterpret_cast<const float4*>(shift + 4 * c4);
        for (int r = rg; r < n; r += RP) {
            float4 v = *reinterpret_cast<const float4*>(Z + (size_t)(row0 + r) * ldz + 4 * c4);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            if (relu) {
                v.x = gnm_relu(v.x); v.y = gnm_relu(v.y); v.z = gnm_relu(v.z); v.w = gnm_relu(v.w);
            }
            if (Hout) *reinterpret_cast<float4*>(Hout + (size_t)(row0 + r) * ldh + 4 * c4) = v;   // (kernel-uniform)
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        red[rg * H4 + c4] = acc;
    }
    if (!pooled) return;
    __syncthreads();
    if (tid < H4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < RP; ++g) {
            const float4 v = red[g * H4 + tid];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (average) {
            const float w = 1.0f / (float)n;    // the reference stores 1./len(graph.g) as fp32 (graphcnn.py:123,130)
            s.x *= w; s.y *= w; s.z *= w; s.w *= w;
        }
        *reinterpret_cast<float4*>(pooled + (size_t)b * ldp + 4 * tid) = s;
    }
}

extern "C" int gnm_bn_relu_readout(const float* Z, int ldz, const float* scale, const float* shift, float* Hout,
                                   int ldh, const int32_t* node_off, int B, int H, int relu, float* pooled, int ldp,
                                   int average, void* stream) {
    if (B <= 0) return GNM_OK;
    if (H <= 0 || (H & 3) || H > 1024 || (ldz & 3) || (Hout && (ldh & 3)) || (pooled && (ldp & 3))) return GNM_ERR_BAD_ARG;
    if (!Hout && !pooled) return GNM_OK;             // nothing to produce
    // one workgroup per graph: with fewer graphs than ~4 per CU, 256-thread workgroups leave the chip idle
    const int threads = per_graph_threads(B);
    const int H4 = H >> 2, RP = threads / H4;
    hipLaunchKernelGGL(gnm_bn_relu_readout_kernel, dim3(B), dim3(threads), (size_t)RP * H4 * 16,
                       reinterpret_cast<hipStream_t>(stream), Z, ldz, scale, shift, Hout, ldh, node_off, H, relu,
                       pooled, ldp, average);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------
// X_concat (graphcnn.py:195): the rows of every graph of the batch, copied from where the arena keeps them
// (base[b] = first row of graph b in src) to their place in the batch (node_off[b]).  One workgroup per graph; a second
// source / destination pair of the same shape rides along (the arena's cached layer-0 aggregate).  Stands in for the
// arange + broadcast-add + two index_select launches torch needed for the same rows.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gnm_gather_graph_rows_kernel(
    const float* __restrict__ src, const float* __restrict__ src2, int lds, int width, const long long* __restrict__ base,
    const int32_t* __restrict__ node_off, float* __restrict__ dst, float* __restrict__ dst2, int ldd) {
    const int b = blockIdx.x;
    const int row0 = node_off[b];
    const int n = node_off[b + 1] - row0;
    const long long s0 = base[b];
    const int total = n * width;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int r = i / width, c = i - r * width;
        const size_t so = (size_t)(s0 + r) * lds + c, dof = (size_t)(row0 + r) * ldd + c;
        dst[dof] = src[so];
        if (src2) dst2[dof] = src2[so];
    }
}

extern "C" int gnm_gather_graph_rows(const float* src, const float* src2, int lds, int width, const long long* base,
                                     const int32_t* node_off, int B, float* dst, float* dst2, int ldd, void* stream) {
    if (B <= 0) return GNM_OK;
    if (!src || !dst || !base || !node_off || width <= 0 || lds < width || ldd < width || (src2 && !dst2)) return GNM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gnm_gather_graph_rows_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, src2,
                       lds, width, base, node_off, dst, dst2, ldd);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------
// backward, pass 1: assemble the gradient arriving at a BatchNorm+ReLU output from
// all of its producers, apply the ReLU mask, and reduce the two BatchNorm sums.
//   total[v,c] = dH[v,c]                                   (next layer's aggregation backward)
//              + w_b * dpool[b,c]                          (readout backward, graphcnn.py:229)
//              + dsc1[v] * U[b,c]                          (discriminator positive branch)
//              + [v < B] s2sum[inv[v]] * U[inv[v],c]       (negative branch: n_f[idx] gathers ROWS
//                                                           perm[g] < B, graphcnn.py:242 quirk)
//   g = total * (z*scale+shift > 0);  partial[b] = (sum g, sum g*xhat) per column
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) gnm_bn_relu_bwd_stats_kernel(
    const float* __restrict__ dH, int lddh, const float* __restrict__ dpool, int ldp, int average,
    const float* __restrict__ dsc1, const float* __restrict__ U, int ldu, const int32_t* __restrict__ inv_perm,
    const float* __restrict__ s2sum, const float* __restrict__ Z, int ldz, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd, int relu,
    float* __restrict__ G, int ldg, const int32_t* __restrict__ node_off, int B, int H,
    double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* red = reinterpret_cast<float4*>(smem);   // [2][RP][H4]
    const int b = blockIdx.x;
    const int row0 = node_off[b];
    const int n = node_off[b + 1] - row0;
    const int H4 = H >> 2;
    const int RP = (int)blockDim.x / H4;
    const int tid = threadIdx.x;
    const int rg = tid / H4, c4 = tid - rg * H4;
    float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
    if (rg < RP) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * c4);
        const float4 sh = *reinterpret_cast<const float4*>(shift + 4 * c4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * c4);
        const float4 rs = *reinterpret_cast<const float4*>(rstd + 4 * c4);
        float4 pb = make_float4(0.f, 0.f, 0.f, 0.f), ub = pb;
        if (dpool) {
            pb = *reinterpret_cast<const float4*>(dpool + (size_t)b * ldp + 4 * c4);
            if (average) {
                const float w = 1.0f / (float)n;
                pb.x *= w; pb.y *= w; pb.z *= w; pb.w *= w;
            }
        }
        if (dsc1) ub = *reinterpret_cast<const float4*>(U + (size_t)b * ldu + 4 * c4);
        auto one = [&](int v, float4 t, const float4 z, float s) {
            if (dsc1) {
                t.x += s * ub.x; t.y += s * ub.y; t.z += s * ub.z; t.w += s * ub.w;
                if (v < B) {
                    const int gq = gnm_perm_entry(inv_perm[v], B);
                    const float s2 = s2sum[gq];
                    const float4 uq = *reinterpret_cast<const float4*>(U + (size_t)gq * ldu + 4 * c4);
                    t.x += s2 * uq.x; t.y += s2 * uq.y; t.z += s2 * uq.z; t.w += s2 * uq.w;
                }
            }
            if (relu) {
                if (!(z.x * sc.x + sh.x > 0.f)) t.x = 0.f;
                if (!(z.y * sc.y + sh.y > 0.f)) t.y = 0.f;
                if (!(z.z * sc.z + sh.z > 0.f)) t.z = 0.f;
                if (!(z.w * sc.w + sh.w > 0.f)) t.w = 0.f;
            }
            *reinterpret_cast<float4*>(G + (size_t)v * ldg + 4 * c4) = t;
            a1.x += t.x; a1.y += t.y; a1.z += t.z; a1.w += t.w;
            a2.x += t.x * ((z.x - mu.x) * rs.x); a2.y += t.y * ((z.y - mu.y) * rs.y);
            a2.z += t.z * ((z.z - mu.z) * rs.z); a2.w += t.w * ((z.w - mu.w) * rs.w);
        };
        constexpr int UR = 4;                       // rows per thread with all their loads in flight
        int r = rg;
        for (; r + (UR - 1) * RP < n; r += UR * RP) {
            float4 zz[UR], dd[UR];
            float ss[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int v = row0 + r + u * RP;
                zz[u] = *reinterpret_cast<const float4*>(Z + (size_t)v * ldz + 4 * c4);
                dd[u] = dH ? *reinterpret_cast<const float4*>(dH + (size_t)v * lddh + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
                ss[u] = dsc1 ? dsc1[v] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                float4 t = pb;
                t.x += dd[u].x; t.y += dd[u].y; t.z += dd[u].z; t.w += dd[u].w;
                one(row0 + r + u * RP, t, zz[u], ss[u]);
            }
        }
        for (; r < n; r += RP) {
            const int v = row0 + r;
            float4 t = pb;
            if (dH) {
                const float4 d = *reinterpret_cast<const float4*>(dH + (size_t)v * lddh + 4 * c4);
                t.x += d.x; t.y += d.y; t.z += d.z; t.w += d.w;
            }
            one(v, t, *reinterpret_cast<const float4*>(Z + (size_t)v * ldz + 4 * c4), dsc1 ? dsc1[v] : 0.f);
        }
        red[rg * H4 + c4] = a1;
        red[(RP + rg) * H4 + c4] = a2;
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * H; idx += (int)blockDim.x) {
        const int which = idx / H, c = idx - which * H;
        const float* base = reinterpret_cast<const float*>(red + (size_t)which * RP * H4);
        double s = 0.0;
        for (int g = 0; g < RP; ++g) s += (double)base[g * H + c];
        partial[((size_t)b * 2 + which) * H + c] = s;
    }
}

extern "C" int gnm_bn_relu_bwd_stats(const float* dH, int lddh, const float* dpool, int ldp, int average,
                                     const float* dsc1, const float* U, int ldu, const int32_t* inv_perm,
                                     const float* s2sum, const float* Z, int ldz, const float* scale,
                                     const float* shift, const float* mean, const float* rstd, int relu, float* G,
                                     int ldg, const int32_t* node_off, int B, int H, double* partial, void* stream) {
    if (B <= 0) return GNM_OK;
    if (H <= 0 || (H & 3) || H > 128 || (ldz & 3) || (ldg & 3) || (dH && (lddh & 3)) || (dpool && (ldp & 3)) ||
        (dsc1 && (ldu & 3)))
        return GNM_ERR_BAD_ARG;
    const int threads = per_graph_threads(B);
    const int H4 = H >> 2, RP = threads / H4;
    hipLaunchKernelGGL(gnm_bn_relu_bwd_stats_kernel, dim3(B), dim3(threads), (size_t)2 * RP * H4 * 16,
                       reinterpret_cast<hipStream_t>(stream), dH, lddh, dpool, ldp, average, dsc1, U, ldu, inv_perm,
                       s2sum, Z, ldz, scale, shift, mean, rstd, relu, G, ldg, node_off, B, H, partial);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// backward finalize: dgamma = sum g*xhat, dbeta = sum g, and the coefficients of
//   dz = cA * (g - m1 - xhat * m2),  cA = gamma*rstd, m1 = sum g / N, m2 = sum g xhat / N
// (eval mode: m1 = m2 = 0, i.e. dz = g * gamma * rstd).
__global__ void __launch_bounds__(1024) gnm_bn_bwd_finalize_kernel(
    const double* __restrict__ partial, int nblk, int H, long long nrows, const float* __restrict__ gamma,
    const float* __restrict__ rstd, int training, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ cA, float* __restrict__ m1, float* __restrict__ m2) {
    __shared__ double lds[1024];
    __shared__ double tot[2 * kFinCols];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * kFinCols;
    reduce_partials_slice(partial, nblk, H, c0, lds, tot);
    const int c = c0 + tid;
    if (tid < kFinCols && c < H) {
        const double s1 = tot[tid], s2 = tot[kFinCols + tid];
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
        cA[c] = gamma[c] * rstd[c];
        m1[c] = training ? (float)(s1 / (double)nrows) : 0.f;
        m2[c] = training ? (float)(s2 / (double)nrows) : 0.f;
    }
}

extern "C" int gnm_bn_bwd_finalize(const double* partial, int nblk, int H, long long nrows, const float* gamma,
                                   const float* rstd, int training, float* dgamma, float* dbeta, float* cA,
                                   float* m1, float* m2, void* stream) {
    if (H <= 0 || H > 128) return GNM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gnm_bn_bwd_finalize_kernel, dim3((H + kFinCols - 1) / kFinCols), dim3(1024), 0,
                       reinterpret_cast<hipStream_t>(stream), partial, nblk, H, nrows, gamma, rstd, training, dgamma, dbeta, cA, m1, m2);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// backward, pass 2 (elementwise, in place allowed): dz = cA * (g - m1 - xhat*m2)
__global__ void __launch_bounds__(256) gnm_bn_bwd_apply_kernel(
    const float* __restrict__ G, int ldg, const float* __restrict__ Z, int ldz, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ cA, const float* __restrict__ m1,
    const float* __restrict__ m2, float* __restrict__ dZ, int ldd, long long N, int H) {
    const int H4 = H >> 2;
    const long long total = N * H4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long v = idx / H4;
        const int c4 = (int)(idx - v * H4);
        const float4 g = *reinterpret_cast<const float4*>(G + (size_t)v * ldg + 4 * c4);
        const float4 z = *reinterpret_cast<const float4*>(Z + (size_t)v * ldz + 4 * c4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * c4);
        const float4 rs = *reinterpret_cast<const float4*>(rstd + 4 * c4);
        const float4 ca = *reinterpret_cast<const float4*>(cA + 4 * c4);
        const float4 a1 = *reinterpret_cast<const float4*>(m1 + 4 * c4);
        const float4 a2 = *reinterpret_cast<const float4*>(m2 + 4 * c4);
        float4 o;
        o.x = ca.x * (g.x - a1.x - (z.x - mu.x) * rs.x * a2.x);
        o.y = ca.y * (g.y - a1.y - (z.y - mu.y) * rs.y * a2.y);
        o.z = ca.z * (g.z - a1.z - (z.z - mu.z) * rs.z * a2.z);
        o.w = ca.w * (g.w - a1.w - (z.w - mu.w) * rs.w * a2.w);
        *reinterpret_cast<float4*>(dZ + (size_t)v * ldd + 4 * c4) = o;
    }
}

extern "C" int gnm_bn_bwd_apply(const float* G, int ldg, const float* Z, int ldz, const float* mean,
                                const float* rstd, const float* cA, const float* m1, const float* m2, float* dZ,
                                int ldd, long long N, int H, void* stream) {
    if (N <= 0) return GNM_OK;
    if (H <= 0 || (H & 3) || (ldg & 3) || (ldz & 3) || (ldd & 3)) return GNM_ERR_BAD_ARG;
    long long blocks = (N * (H >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gnm_bn_bwd_apply_kernel, dim3((int)blocks), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), G, ldg, Z, ldz, mean, rstd, cA, m1, m2, dZ, ldd, N, H);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// d eps[layer] = sum of the aggregation kernel's fp64 partials (fixed order)
__global__ void __launch_bounds__(1024) gnm_sum_partials_kernel(const double* __restrict__ partial, int count,
                                                                float* __restrict__ out) {
    __shared__ double lds[1024];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < count; i += 1024) s += partial[i];
    lds[tid] = s;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) lds[tid] += lds[tid + off];
        __syncthreads();
    }
    if (tid == 0) *out = (float)lds[0];
}

// d eps[l] when the aggregation backward has no other consumer (layer 0 without input gradients):
// partial[block] = sum over the block's rows of A[r,:] . B[r,:]  (fp64; summed later with the other layers).
static constexpr int kDotBlocks = 512;
__global__ void __launch_bounds__(256) gnm_rowdot_partials_kernel(const float* __restrict__ A, int lda,
                                                                  const float* __restrict__ Bm, int ldb, long long N,
                                                                  int F, double* __restrict__ partial) {
    __shared__ double lds[4];
    const long long total = N * F;
    double acc = 0.0;
    if (lda == F && ldb == F && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bm)) & 15) == 0) {
        // both operands are dense row-major: one flat dot product, 16-B loads
        const long long n4 = total >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(A);
        const float4* b4 = reinterpret_cast<const float4*>(Bm);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const float4 x = a4[i], y = b4[i];
            acc += (double)(x.x * y.x + x.y * y.y) + (double)(x.z * y.z + x.w * y.w);
        }
        if (blockIdx.x == 0 && threadIdx.x < (int)(total & 3)) {
            const long long i = (n4 << 2) + threadIdx.x;
            acc += (double)(A[i] * Bm[i]);
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long r = i / F;
            const int c = (int)(i - r * F);
            acc += (double)(A[r * lda + c] * Bm[r * ldb + c]);
        }
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

extern "C" int gnm_rowdot_num_partials(void) { return kDotBlocks; }

extern "C" int gnm_rowdot_partials(const float* A, int lda, const float* Bm, int ldb, long long N, int F,
                                   double* partial, void* stream) {
    if (N < 0 || F <= 0 || !partial || (N > 0 && (!A || !Bm))) return GNM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gnm_rowdot_partials_kernel, dim3(kDotBlocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       A, lda, Bm, ldb, N, F, partial);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// the same for several independent sets in one launch (the d eps of all layers): set k = partial[k*stride .. +counts[k])
struct SumCounts {
    int n[16];
};
__global__ void __launch_bounds__(1024) gnm_sum_partials_multi_kernel(const double* __restrict__ partial,
                                                                      long long stride, const SumCounts counts,
                                                                      float* __restrict__ out) {
    __shared__ double lds[1024];
    const int tid = threadIdx.x, k = blockIdx.x;
    const double* p = partial + (long long)k * stride;
    const int count = counts.n[k];
    double s = 0.0;
    for (int i = tid; i < count; i += 1024) s += p[i];
    lds[tid] = s;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) lds[tid] += lds[tid + off];
        __syncthreads();
    }
    if (tid == 0) out[k] = (float)lds[0];
}

extern "C" int gnm_sum_partials_multi(const double* partial, long long stride, const int* counts_host, int nsets,
                                      float* out, void* stream) {
    if (nsets <= 0) return GNM_OK;
    if (nsets > 16 || !partial || !counts_host || !out) return GNM_ERR_BAD_ARG;
    SumCounts c;
    for (int k = 0; k < 16; ++k) c.n[k] = k < nsets ? counts_host[k] : 0;
    hipLaunchKernelGGL(gnm_sum_partials_multi_kernel, dim3(nsets), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream),
                       partial, stride, c, out);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

extern "C" int gnm_sum_partials(const double* partial, int count, float* out, void* stream) {
    hipLaunchKernelGGL(gnm_sum_partials_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream),
                       partial, count, out);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}
