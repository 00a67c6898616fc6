// One-launch evaluation encoder (round 3): the L GIN layers of GIN_InfoMaxReg.forward in eval() mode
// (/root/reference models/graphcnn.py:208-231 with BatchNorm on its running statistics, mlp.py:40-49), ONE WORKGROUP PER
// GRAPH, all layers in one kernel.
//
// Why: the reference evaluates one graph per forward (main.py:49-57, over the whole training set every epoch, :154; also
// get_latent_space, :71-82).  Through the training kernels such a forward is a dependent chain of ~110 small launches
// = 0.57 ms of GPU time per 400-node graph even when replayed from a hipGraph -- launch latency, not work.  In eval mode
// BatchNorm is an affine map, nothing couples the graphs of a batch, and a graph's whole layer fits one CU's resources:
// per layer the workgroup (16 waves)
//   1. aggregates: per 32-column block, the input tile -> three bf16 planes in LDS, bit adjacency x planes on MFMA
//      (the product of csrc/aggm.hip), self term / degree division in the epilogue -> `pooled` (global scratch, L2);
//   2. runs the MLP: each Linear as the six-term split-precision bf16 product of csrc/linear.hip (weight planes in LDS,
//      A fragments loaded straight from the L2-resident input), bias + folded BatchNorm + ReLU in the epilogue;
//   3. the last Linear's epilogue also applies the layer's outer BatchNorm + ReLU, writes the hidden layer (the
//      discriminator's score kernel reads it) and reduces the graph readout (sum / mean) in a fixed order.
// Then the classifier head (graphcnn.py:224-231, dropout off) and sigmoid(g_f) for the discriminator (:239).
// Arithmetic is the training kernels' (fp32-faithful three-plane splits, fp32 accumulation); results agree with them to
// fp32 rounding, not bitwise.  The MFMA work of one graph on one CU bounds it at ~57 us (aggregation 33 + Linears 24);
// B graphs run on B CUs in the same launch.
#include "gnm_common.h"
#include <string.h>

typedef __bf16 ev_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int ev_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ev_u32x2 __attribute__((ext_vector_type(2)));

#define GNM_EVAL_MAX_LAYERS 16
#define GNM_EVAL_MAX_MLP 3
static constexpr int kEvThreads = 768;               // 12 waves = 3 per SIMD: 168 registers per lane (the Linear phase
static constexpr int kEvWaves = 12;                   // holds 48 operand + 32 accumulator registers)
static constexpr int kEvMaxN = 400;               // 13 row blocks; planes of one 32-column block: 75 KB
static constexpr int kEvH = 64;
static constexpr unsigned kEvK8 = 512;            // bytes per k-group (8 rows x 32 columns bf16) of a plane
static constexpr unsigned kEvStep = 2 * kEvK8;

// Parameter pointers live in a small DEVICE table (the caller fills it once per model; a by-value struct of 400
// pointers indexed by the layer counter ended up in scratch memory): int64 words,
//   entry j = l * m + k, 7 words: W, bias, gamma, beta, running_mean, running_var, ldw (Linear k of layer l and the
//   BatchNorm behind it), then per layer l 2 words: Wp, bp (classifier Linear).
static constexpr int kEvLinWords = 7;
struct EvalArgs {
    const uint32_t* adj_bits; const int64_t* b_bits_off; const int32_t* node_off;
    const int32_t* rowptr; const int64_t* b_rp_off;      // degrees (neighbour "average")
    const float* X; int ldx, F0;
    int B, n16_max, L, m, C;
    int average, self_loop, graph_avg;
    float bn_eps;
    const long long* table;                              // see above
    const float* eps;                                    // [L] or null (learn_eps False)
    float* hidden; long long hidden_stride; int ldh;     // [L][N, H]
    float* s0; float* s1; int lds_;                      // two [N, H] scratch arrays (pooled / MLP intermediates)
    float* g_f; int ldgf;                                // [B, L * H]
    float* c_sig;                                        // [B, L * H] sigmoid(g_f), or null
    float* c_logit; int ldc;                             // [B, C]
};

__device__ __forceinline__ unsigned ev_pair_hi(unsigned lo_word, unsigned hi_word) {
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);
}
__device__ __forceinline__ void ev_split3(const float f, unsigned& a1, unsigned& a2, unsigned& a3) {
    a1 = __float_as_uint(f) & 0xFFFF0000u;
    const float r1 = f - __uint_as_float(a1);
    a2 = __float_as_uint(r1) & 0xFFFF0000u;
    a3 = __float_as_uint(r1 - __uint_as_float(a2));
}
__device__ __forceinline__ void ev_split8(const float* f, ev_u32x4& p1, ev_u32x4& p2, ev_u32x4& p3) {
    unsigned a1[8], a2[8], a3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ev_split3(f[j], a1[j], a2[j], a3[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p1[j] = ev_pair_hi(a1[2 * j], a1[2 * j + 1]);
        p2[j] = ev_pair_hi(a2[2 * j], a2[2 * j + 1]);
        p3[j] = ev_pair_hi(a3[2 * j], a3[2 * j + 1]);
    }
}
__host__ __device__ static inline int ev_half_words(int W) { return (((W + 1) >> 1) + 3) & ~3; }

__global__ void __launch_bounds__(kEvThreads) gnm_eval_encoder_kernel(const EvalArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int H = kEvH, HT = 2;
    constexpr int E = 4 * HT * 64;                     // 16-byte operand entries per weight plane ([m][c][lane])
    const unsigned plane_bytes = (unsigned)(p.n16_max >> 3) * kEvK8;
    char* planes = smem;                                                   // [3][n16 / 8][512]
    ev_u32x4* Wpl = reinterpret_cast<ev_u32x4*>(smem + 3u * plane_bytes);  // [3][E] weight planes of the current Linear
    char* lut = reinterpret_cast<char*>(Wpl + 3 * E);                      // 128 B nibble table
    float* bnv = reinterpret_cast<float*>(lut + 128);                      // [3][H]: bias, scale, shift of the current Linear
    float* rsum = bnv + 3 * H;                                             // [16 waves][H] readout partials
    float* gfl = rsum + kEvWaves * H;                                      // [GNM_EVAL_MAX_LAYERS * H] this graph's g_f

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int row0 = p.node_off[b];
    const int n = p.node_off[b + 1] - row0;
    const int LH = p.L * H;
    if (n <= 0) {              // an empty graph: empty readout (0, or 0/0 for the mean), logits from the biases alone
        for (int e = tid; e < LH; e += kEvThreads) {
            const float g = p.graph_avg ? 0.f / 0.f : 0.f;
            p.g_f[(size_t)b * p.ldgf + e] = g;
            if (p.c_sig) p.c_sig[(size_t)b * p.ldgf + e] = 1.f / (1.f + expf(-g));
        }
        if (tid < p.C) {
            float acc = 0.f;
            for (int ll = 0; ll < p.L; ++ll) {
                const float* bpl = reinterpret_cast<const float*>(p.table[(size_t)p.L * p.m * kEvLinWords + 2 * ll + 1]);
                acc += bpl[tid] + (p.graph_avg ? 0.f / 0.f : 0.f);
            }
            p.c_logit[(size_t)b * p.ldc + tid] = acc;
        }
        return;
    }
    const int W = (n + 31) >> 5;
    const int ksteps = (n + 15) >> 4;
    const int n16 = ksteps * 16;
    const int HPW = ev_half_words(W);
    const uint32_t* gbits = p.adj_bits + p.b_bits_off[b];
    const int32_t* rp = p.rowptr + p.b_rp_off[b];

    if (tid < 16) {            // nibble e -> bf16 (bit 0, bit 1, bit 2, bit 3) as two words
        const unsigned one = 0x3F80u;
        ev_u32x2 v;
        v.x = ((tid & 1) ? one : 0u) | ((tid & 2) ? one << 16 : 0u);
        v.y = ((tid & 4) ? one : 0u) | ((tid & 8) ? one << 16 : 0u);
        *reinterpret_cast<ev_u32x2*>(lut + 8 * tid) = v;
    }
    // a wave's rows of the bit adjacency: row block `wave` is kept in registers for the whole kernel (the same for
    // every layer and column block); a wave that also owns block wave + 12 loads that one when it gets there
    auto load_pk = [&](int rb, unsigned (&pkv)[8]) {
        const ev_u32x4* ra = reinterpret_cast<const ev_u32x4*>(gbits + (size_t)(min(rb, W - 1) * 32 + i) * (2 * HPW) + h * HPW);
        const ev_u32x4 z4 = {0u, 0u, 0u, 0u};
        const ev_u32x4 q0 = ra[0], q1 = HPW > 4 ? ra[1] : z4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pkv[j] = q0[j]; pkv[4 + j] = q1[j]; }
    };
    unsigned pk0[8];
    load_pk(wave, pk0);

    for (int l = 0; l < p.L; ++l) {
        const int F = l == 0 ? p.F0 : H;
        const float* xin = l == 0 ? p.X : p.hidden + (size_t)(l - 1) * p.hidden_stride;
        const int ldin = l == 0 ? p.ldx : p.ldh;
        const float selfB = p.self_loop ? 0.f : (p.eps ? 1.f + p.eps[l] : 1.f);
        // ================= 1. aggregation, one 32-column block at a time -> s0 [n, F] =========================
        for (int col0 = 0; col0 < F; col0 += 32) {
            __syncthreads();               // the planes are free (previous block's epilogue / previous layer done)
            // tile -> planes: item = (row quad rq, 4-column chunk c4); columns >= F are zero
            for (int it = tid; it < (n16 >> 2) * 8; it += kEvThreads) {
                const int c4 = it & 7, rq = it >> 3;
                float v[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = 4 * rq + k;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int col = col0 + 4 * c4 + c;
                        v[k][c] = (row < n && col < F) ? xin[(size_t)(row0 + row) * ldin + col] : 0.f;
                    }
                }
                const unsigned off = (unsigned)(rq >> 1) * kEvK8 + (unsigned)((4 * c4 * 8 + 4 * (rq & 1)) * 2);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned x0[4], x1[4], x2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ev_split3(v[k][c], x0[k], x1[k], x2[k]);
                    ev_u32x2 w0, w1, w2;
                    w0.x = ev_pair_hi(x0[0], x0[1]); w0.y = ev_pair_hi(x0[2], x0[3]);
                    w1.x = ev_pair_hi(x1[0], x1[1]); w1.y = ev_pair_hi(x1[2], x1[3]);
                    w2.x = ev_pair_hi(x2[0], x2[1]); w2.y = ev_pair_hi(x2[2], x2[3]);
                    char* dst = planes + off + c * 16;
                    *reinterpret_cast<ev_u32x2*>(dst) = w0;
                    *reinterpret_cast<ev_u32x2*>(dst + plane_bytes) = w1;
                    *reinterpret_cast<ev_u32x2*>(dst + 2u * plane_bytes) = w2;
                }
            }
            __syncthreads();
            for (int rb = wave; rb < W; rb += kEvWaves) {
                unsigned pk[8];
                if (rb == wave) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) pk[j] = pk0[j];
                } else {
                    load_pk(rb, pk);
                }
                const char* bp0 = planes + h * kEvK8 + i * 16;
                const char* bp1 = bp0 + plane_bytes;
                const char* bp2 = bp1 + plane_bytes;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                auto afrag = [&](unsigned pkw, int mm) -> ev_bf16x8 {
                    const unsigned byte3 = mm == 0 ? (pkw << 3) : (pkw >> (8 * mm - 3));
                    const unsigned lo = byte3 & 0x78u, hi = (byte3 >> 4) & 0x78u;
                    const ev_u32x2 l2 = *reinterpret_cast<const ev_u32x2*>(lut + lo);
                    const ev_u32x2 h2 = *reinterpret_cast<const ev_u32x2*>(lut + hi);
                    const ev_u32x4 q = {l2.x, l2.y, h2.x, h2.y};
                    return __builtin_bit_cast(ev_bf16x8, q);
                };
                auto bfrag = [&](const char* bp, int ks) -> ev_bf16x8 {
                    return __builtin_bit_cast(ev_bf16x8, *reinterpret_cast<const ev_u32x4*>(bp + ks * kEvStep));
                };
                ev_bf16x8 b0 = bfrag(bp0, 0), b1 = bfrag(bp1, 0), b2 = bfrag(bp2, 0);
                ev_bf16x8 aA = afrag(pk[0], 0);
#pragma unroll
                for (int ks = 0; ks < 25; ++ks) {
                    if (ks < ksteps) {                                // wave-uniform
                        constexpr int LASTK = 24;
                        const int kn = ks < LASTK ? ks + 1 : LASTK;
                        const ev_bf16x8 nA = afrag(pk[kn >> 2], kn & 3);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b0, acc, 0, 0, 0);
                        b0 = bfrag(bp0, kn);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b1, acc, 0, 0, 0);
                        b1 = bfrag(bp1, kn);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aA, b2, acc, 0, 0, 0);
                        b2 = bfrag(bp2, kn);
                        aA = nA;
                    }
                }
                // epilogue: lane = column, 16 rows per lane (r = 4 k + q: row rb * 32 + 8 k + 4 h + q)
                const int col = col0 + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int vrow = rb * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    const char* e = planes + (unsigned)(min(vrow, n16 - 1) >> 3) * kEvK8 + i * 16 + (vrow & 7) * 2;
                    const float e1 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e) << 16);
                    const float e2 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e + plane_bytes) << 16);
                    const float e3 = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(e + 2u * plane_bytes) << 16);
                    const float own = (e1 + e2) + e3;              // the tile's own value: the planes add up to it exactly
                    float tot = acc[r];
                    if (p.self_loop) tot += own;
                    if (p.average) {
                        const int vc = min(vrow, n - 1);
                        tot /= (float)(rp[vc + 1] - rp[vc] + p.self_loop);      // 0/0 -> NaN as the reference
                    }
                    if (!p.self_loop) tot += selfB * own;
                    if (vrow < n && col < F) p.s0[(size_t)(row0 + vrow) * p.lds_ + col] = tot;
                }
            }
        }
        // ================= 2. the MLP (mlp.py:40-49) + the layer's outer BatchNorm + ReLU + readout ===========
        const float* lin_in = p.s0;
        int K = F;
        for (int k = 0; k < p.m; ++k) {
            const bool last = k == p.m - 1;
            const long long* te = p.table + (size_t)(l * p.m + k) * kEvLinWords;
            const float* Wg = reinterpret_cast<const float*>(te[0]);
            const float* bias_g = reinterpret_cast<const float*>(te[1]);
            const float* gamma_g = reinterpret_cast<const float*>(te[2]);
            const float* beta_g = reinterpret_cast<const float*>(te[3]);
            const float* rmean_g = reinterpret_cast<const float*>(te[4]);
            const float* rvar_g = reinterpret_cast<const float*>(te[5]);
            const int ldw = (int)te[6];
            float* lin_out = last ? p.hidden + (size_t)l * p.hidden_stride : ((k & 1) ? p.s0 : p.s1);
            const int ld_in = (k == 0 || (k & 1) == 0) ? p.lds_ : p.lds_;   // both scratch arrays share a leading dimension
            const int ld_out = last ? p.ldh : p.lds_;
            __syncthreads();           // the previous phase's global writes are visible; weight planes / bnv are free
            // weight planes: entry (mm, c, lane = 32 kg + nn): W[h = 32 c + nn][k = 8 mm + 32 kg + 0..7], zero past K
            {
                for (int e = tid; e < E; e += kEvThreads) {
                    const int nn = e & 31, kg = (e >> 5) & 1, c = (e >> 6) % HT, mm = e / (64 * HT);
                    float f[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kk = 8 * mm + 32 * kg + j;
                        f[j] = kk < K ? Wg[(size_t)(32 * c + nn) * ldw + kk] : 0.f;
                    }
                    ev_u32x4 p1, p2, p3;
                    ev_split8(f, p1, p2, p3);
                    Wpl[e] = p1; Wpl[E + e] = p2; Wpl[2 * E + e] = p3;
                }
                if (tid < H) {         // bias and the folded eval-mode BatchNorm behind this Linear (running statistics)
                    const float sc = gamma_g[tid] / sqrtf(rvar_g[tid] + p.bn_eps);
                    bnv[tid] = bias_g ? bias_g[tid] : 0.f;
                    bnv[H + tid] = sc;
                    bnv[2 * H + tid] = beta_g[tid] - rmean_g[tid] * sc;
                }
            }
            __syncthreads();
            float cs[HT] = {0.f, 0.f};                 // readout partials of this lane's columns (last Linear only)
            for (int t = wave; t < W; t += kEvWaves) {         // 32-row tiles (13 at n = 400: one per wave)
                const int r0 = t * 32;
                const int arow = min(r0 + i, n - 1);
                // A fragments straight from the (L2-resident) input: row i, k = 8 mm + 32 h + 0..7
                ev_u32x4 A1[4], A2[4], A3[4];
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    float f[8];
                    const int k0 = 8 * mm + 32 * h;
                    const float* src = lin_in + (size_t)(row0 + arow) * ld_in + k0;
                    if (k0 + 8 <= K && (ld_in & 3) == 0) {
                        const float4 v0 = *reinterpret_cast<const float4*>(src);
                        const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
                        f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] = k0 + j < K ? src[j] : 0.f;
                    }
                    ev_split8(f, A1[mm], A2[mm], A3[mm]);
                }
                f32x16 acc[HT];
#pragma unroll
                for (int c = 0; c < HT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    if (8 * mm < K || 32 + 8 * mm < K) {       // wave-uniform: k groups past K are all zero
                        const ev_bf16x8 a1 = __builtin_bit_cast(ev_bf16x8, A1[mm]), a2 = __builtin_bit_cast(ev_bf16x8, A2[mm]),
                                        a3 = __builtin_bit_cast(ev_bf16x8, A3[mm]);
#pragma unroll
                        for (int c = 0; c < HT; ++c) {
                            const int e = (mm * HT + c) * 64 + lane;
                            const ev_bf16x8 w1 = __builtin_bit_cast(ev_bf16x8, Wpl[e]), w2 = __builtin_bit_cast(ev_bf16x8, Wpl[E + e]),
                                            w3 = __builtin_bit_cast(ev_bf16x8, Wpl[2 * E + e]);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w3, acc[c], 0, 0, 0);      // small terms first
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, w1, acc[c], 0, 0, 0);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w2, acc[c], 0, 0, 0);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w2, acc[c], 0, 0, 0);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w1, acc[c], 0, 0, 0);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w1, acc[c], 0, 0, 0);
                        }
                    }
                }
                // epilogue: z = acc + bias; BatchNorm (affine in eval mode) + ReLU; store; readout sums on the last Linear
#pragma unroll
                for (int c = 0; c < HT; ++c) {
                    const int col = 32 * c + i;
                    const float bz = bnv[col], sc = bnv[H + col], sh = bnv[2 * H + col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float z = acc[c][r] + bz;
                        const float a = gnm_relu(z * sc + sh);
                        if (r0 + lrow < n) {
                            lin_out[(size_t)(row0 + r0 + lrow) * ld_out + col] = a;
                            cs[c] += a;
                        }
                    }
                }
            }
            if (last) {                // graph readout of the layer (graphcnn.py:229): waves in a fixed order
#pragma unroll
                for (int c = 0; c < HT; ++c) {
                    cs[c] += __shfl_xor(cs[c], 32, 64);
                    if (h == 0) rsum[wave * H + 32 * c + i] = cs[c];
                }
                __syncthreads();
                if (tid < H) {
                    float s = 0.f;
                    for (int w = 0; w < kEvWaves; ++w) s += rsum[w * H + tid];
                    if (p.graph_avg) s *= 1.0f / (float)n;
                    gfl[l * H + tid] = s;
                    p.g_f[(size_t)b * p.ldgf + l * H + tid] = s;
                    if (p.c_sig) p.c_sig[(size_t)b * p.ldgf + l * H + tid] = 1.f / (1.f + expf(-s));
                }
            }
            lin_in = lin_out;
            K = H;
        }
    }
    // ================= 3. classifier head (graphcnn.py:224-231, eval: no dropout) ==============================
    __syncthreads();
    if (tid < p.C) {
        float acc = 0.f;
        for (int l = 0; l < p.L; ++l) {
            const long long* th = p.table + (size_t)p.L * p.m * kEvLinWords + 2 * l;
            float z = reinterpret_cast<const float*>(th[1])[tid];
            const float* w = reinterpret_cast<const float*>(th[0]) + (size_t)tid * H;
            for (int c = 0; c < H; ++c) z += gfl[l * H + c] * w[c];
            acc += z;
        }
        p.c_logit[(size_t)b * p.ldc + tid] = acc;
    }
}

extern "C" int gnm_eval_max_nodes(void) { return kEvMaxN; }

extern "C" long long gnm_eval_table_words(int L, int m) { return (long long)L * m * kEvLinWords + 2LL * L; }

// One launch = the eval-mode encoder + readout + classifier of B graphs (see the file header).  `table`: DEVICE array of
// gnm_eval_table_words(L, m) int64 words holding the parameter pointers -- entry l * m + k (7 words: W, bias, gamma, beta,
// running_mean, running_var as device addresses, then the weight's leading dimension) for Linear k of layer l's MLP and
// the BatchNorm BEHIND it (the MLP's inner BatchNorm k for k < m - 1, the layer's outer BatchNorm for k = m - 1), then
// per layer 2 words (classifier weight [C, H] row-major, bias [C]).  hidden: [L][N, H] (layer stride hidden_stride
// floats, leading dimension ldh).  H must be 64, 1 <= m <= 3, L <= 16, F0 <= 64, C <= 64, every graph needs a bit
// adjacency and at most gnm_eval_max_nodes() nodes: GNM_ERR_UNSUPPORTED otherwise (the caller then runs the
// layer-by-layer path).  s0 / s1: two [N, lds] fp32 scratch arrays.  c_sig may be NULL.
extern "C" int gnm_eval_encoder(const uint32_t* adj_bits, const int64_t* b_bits_off, const int32_t* node_off,
                                const int32_t* rowptr, const int64_t* b_rp_off, int B, int n_max, const float* X, int ldx,
                                int F0, int H, int L, int m, int C, int average, int self_loop, int graph_avg,
                                float bn_eps, const long long* table, const float* eps, float* hidden,
                                long long hidden_stride, int ldh, float* s0, float* s1, int lds_, float* g_f, int ldgf,
                                float* c_sig, float* c_logit, int ldc, void* stream) {
    if (B <= 0) return GNM_OK;
    if (H != kEvH || m < 1 || m > GNM_EVAL_MAX_MLP || L < 1 || L > GNM_EVAL_MAX_LAYERS || F0 < 1 || F0 > 64 || C < 1 ||
        C > 64 || n_max < 1 || n_max > kEvMaxN)
        return GNM_ERR_UNSUPPORTED;
    if (!adj_bits || !b_bits_off || !node_off || !rowptr || !b_rp_off || !X || !table || !hidden || !s0 || !s1 || !g_f ||
        !c_logit)
        return GNM_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(adj_bits) & 15) || (ldh & 3) || (lds_ & 3)) return GNM_ERR_UNSUPPORTED;
    EvalArgs a;
    memset(&a, 0, sizeof(a));
    a.adj_bits = adj_bits; a.b_bits_off = b_bits_off; a.node_off = node_off; a.rowptr = rowptr; a.b_rp_off = b_rp_off;
    a.X = X; a.ldx = ldx; a.F0 = F0; a.B = B; a.n16_max = ((n_max + 15) / 16) * 16; a.L = L; a.m = m; a.C = C;
    a.average = average; a.self_loop = self_loop; a.graph_avg = graph_avg; a.bn_eps = bn_eps; a.eps = eps;
    a.table = table; a.hidden = hidden; a.hidden_stride = hidden_stride;
    a.ldh = ldh; a.s0 = s0; a.s1 = s1; a.lds_ = lds_; a.g_f = g_f; a.ldgf = ldgf; a.c_sig = c_sig; a.c_logit = c_logit;
    a.ldc = ldc;
    const size_t lds = (size_t)3 * (a.n16_max / 8) * kEvK8 + (size_t)3 * 4 * 2 * 64 * 16 + 128 + (size_t)3 * kEvH * 4 +
                       (size_t)kEvWaves * kEvH * 4 + (size_t)GNM_EVAL_MAX_LAYERS * kEvH * 4;
    if (lds > (size_t)kLdsBudget) return GNM_ERR_UNSUPPORTED;
    GNM_ALLOW_FULL_LDS(&gnm_eval_encoder_kernel);
    hipLaunchKernelGGL(gnm_eval_encoder_kernel, dim3(B), dim3(kEvThreads), lds, reinterpret_cast<hipStream_t>(stream), a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}
