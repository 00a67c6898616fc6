// Evaluation encoder, one launch PER LAYER with one workgroup per 32-row block of a graph (round 3): the L GIN layers
// of GIN_InfoMaxReg.forward in eval() mode (/root/reference models/graphcnn.py:208-231 with BatchNorm on its running
// statistics, mlp.py:40-49) for the reference's evaluation pattern -- one graph per forward (main.py:49-57, :71-82).
//
// Why a third form: replayed through the training kernels such a forward is ~100 dependent launches (~190 us of GPU time
// per 400-node graph); the one-workgroup-per-graph encoder (evalfwd.hip) is one launch but runs the whole graph's matrix
// work on ONE CU (~200 us).  Here a layer is one launch whose grid is (graph, 32-row block): 13 workgroups per 400-node
// graph run on 13 CUs, and the only thing a row block needs from the others is the previous layer's activations -- which
// the launch boundary provides.  Per workgroup (4 waves):
//   A. aggregation of its 32 output rows, operand roles swapped (Y^T = H^T x Adj^T): the activations are the A operand,
//      loaded straight from global memory as "eight consecutive rows of one column per lane" (4-byte loads with
//      lane = column), split in registers into three exact bf16 planes; the block's adjacency bits are the B operand
//      (expanded through a 16-entry LDS table, as csrc/aggm.hip does).  The waves split (column tile, k range); partial
//      tiles meet in LDS, where the self term / degree division are applied.
//   B. the MLP on the 32 x F tile in LDS: each Linear as the six-term split-precision product (csrc/linear.hip), A
//      fragments from the LDS tile, W rows straight from global memory, bias + folded BatchNorm + ReLU on the way back
//      into LDS; the last Linear's epilogue applies the layer's outer BatchNorm + ReLU, writes the block's rows of the
//      hidden layer and its share of the graph readout (fixed order).
// A last small launch adds the readout shares, applies the classifier head (graphcnn.py:224-231, dropout off) and
// sigmoid(g_f) (:239).  Arithmetic is fp32-faithful (three-plane splits, fp32 accumulation): results agree with the
// training kernels to fp32 rounding, not bitwise.
#include "gnm_common.h"
#include <string.h>

typedef __bf16 el_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int el_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int el_u32x2 __attribute__((ext_vector_type(2)));

static constexpr int kElMaxN = 416;               // 13 row blocks, 26 steps: the bit rows of a block live in 8 registers
static constexpr int kElMaxH = 128;
static constexpr int kElLinWords = 7;             // the parameter table of evalfwd.hip (gnm_eval_table_words)
static constexpr int kElTS = kElMaxH + 4;         // row stride of the LDS tiles (floats)

struct ElArgs {
    const uint32_t* adj_bits; const int64_t* b_bits_off; const int32_t* node_off;
    const int32_t* rowptr; const int64_t* b_rp_off;
    const float* Hin; int ldin, Fin;
    int B, wmax, L, m, l, H;
    int average, self_loop;
    float bn_eps;
    const float* eps;                             // [L] on the device, or null (learn_eps False)
    const long long* table;
    float* Hout; int ldh;
    float* rpart;                                 // [B][wmax][H]: this layer's readout shares
};

__device__ __forceinline__ void el_split8(const float* f, el_bf16x8& p1, el_bf16x8& p2, el_bf16x8& p3) {
    unsigned a1[8], a2[8], a3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a1[j] = __float_as_uint(f[j]) & 0xFFFF0000u;
        const float r1 = f[j] - __uint_as_float(a1[j]);
        a2[j] = __float_as_uint(r1) & 0xFFFF0000u;
        a3[j] = __float_as_uint(r1 - __uint_as_float(a2[j]));
    }
    el_u32x4 q1, q2, q3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q1[j] = __builtin_amdgcn_perm(a1[2 * j + 1], a1[2 * j], 0x07060302u);
        q2[j] = __builtin_amdgcn_perm(a2[2 * j + 1], a2[2 * j], 0x07060302u);
        q3[j] = __builtin_amdgcn_perm(a3[2 * j + 1], a3[2 * j], 0x07060302u);
    }
    p1 = __builtin_bit_cast(el_bf16x8, q1); p2 = __builtin_bit_cast(el_bf16x8, q2); p3 = __builtin_bit_cast(el_bf16x8, q3);
}

__global__ void __launch_bounds__(256) gnm_eval_layer_kernel(const ElArgs p) {
    __shared__ __attribute__((aligned(16))) float T0[32 * kElTS];
    __shared__ __attribute__((aligned(16))) float T1[32 * kElTS];
    __shared__ __attribute__((aligned(16))) float part[4][32][33];
    __shared__ __attribute__((aligned(16))) char lut[128];
    __shared__ unsigned bitsw[8][256];            // word j of thread t's half row of the block's adjacency bits
    __shared__ float aff[3][3][kElMaxH];          // per Linear of the MLP: bias, scale, shift (the BatchNorm behind it, folded)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / p.wmax, rb = blockIdx.x - b * p.wmax;
    const int row0 = p.node_off[b];
    const int n = p.node_off[b + 1] - row0;
    const int W = (n + 31) >> 5;
    if (rb >= W) return;                          // (also an empty graph: no rows, no readout share)
    const int H = p.H, Fin = p.Fin;
    const int ksteps = (n + 15) >> 4;
    if (tid < 16) {            // nibble e -> bf16 (bit 0, bit 1, bit 2, bit 3) as two words
        const unsigned one = 0x3F80u;
        el_u32x2 v;
        v.x = ((tid & 1) ? one : 0u) | ((tid & 2) ? one << 16 : 0u);
        v.y = ((tid & 4) ? one : 0u) | ((tid & 8) ? one << 16 : 0u);
        *reinterpret_cast<el_u32x2*>(lut + 8 * tid) = v;
    }
    // ---- everything the MLP needs that does not depend on the tile, requested NOW: the parameter table -> the
    //      pointers -> the vectors and this wave's first two W fragments per Linear are three dependent round trips to
    //      memory per Linear; taken one Linear at a time behind the aggregation they were most of this kernel's 17 us
    const int NCT = H >> 5, KSB = 4 / NCT;
    const int ctB = wave % NCT, khB = wave / NCT;
    const int ncolB = 32 * ctB + i;                               // output column of this lane = row of W
    const float* Wk[3];
    int ldwk[3];
    float fbw[3][2][8];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Wk[k] = nullptr; ldwk[k] = 0;
        if (k < p.m) {
            const long long* te = p.table + (size_t)(p.l * p.m + k) * kElLinWords;
            Wk[k] = reinterpret_cast<const float*>(te[0]);
            ldwk[k] = (int)te[6];
            const int K = k == 0 ? Fin : H;
            if (tid < H) {
                const float gam = reinterpret_cast<const float*>(te[2])[tid], bet = reinterpret_cast<const float*>(te[3])[tid];
                const float rm = reinterpret_cast<const float*>(te[4])[tid], rv = reinterpret_cast<const float*>(te[5])[tid];
                const float rstd = (float)(1.0 / sqrt((double)rv + (double)p.bn_eps));
                const float sc = gam * rstd;
                aff[k][0][tid] = reinterpret_cast<const float*>(te[1])[tid];
                aff[k][1][tid] = sc;
                aff[k][2][tid] = bet - rm * sc;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k0 = 16 * (khB + KSB * u) + 8 * h;
#pragma unroll
                for (int j = 0; j < 8; ++j) fbw[k][u][j] = k0 + j < K ? Wk[k][(size_t)ncolB * ldwk[k] + k0 + j] : 0.f;
            }
        }
    }
    // ---- A. aggregation ----------------------------------------------------------------------------------------
    const int NCA = Fin <= 32 ? 1 : (Fin <= 64 ? 2 : 4);          // column tiles of the input; the rest of the waves split k
    // the combine pass's own operands (8 threads per tile row): this thread's elements of the self term and its row's
    // degree -- requested now, used after the product (fetched inside the combine loop they were eight dependent
    // round trips to L2)
    const int row = tid >> 3, c8 = tid & 7;
    const int grow = row0 + min(rb * 32 + row, n - 1);
    const bool vrow = rb * 32 + row < n;
    // (parked in the second LDS tile, which the MLP does not touch before the combine pass has read it)
    for (int c = c8; c < NCA * 32; c += 8)
        T1[row * kElTS + c] = c < Fin ? p.Hin[(size_t)grow * p.ldin + c] : 0.f;
    float deg = 1.f;
    if (p.average) {
        const int32_t* rp = p.rowptr + p.b_rp_off[b];
        const int vr = min(rb * 32 + row, n - 1);
        deg = (float)(rp[vr + 1] - rp[vr] + p.self_loop);
    }
    {
        const int ct = wave % NCA, kh = wave / NCA, KS = 4 / NCA;
        const int HPW = (((W + 1) >> 1) + 3) & ~3;
        {   // this lane's half row of the block's adjacency bits -> LDS, one word per (word index, thread): the step loop
            // below is a real loop (fully unrolled, 26 steps x 3 Linears were 54 KB of straight-line code that every
            // workgroup ran once from a cold instruction cache: 17 us per launch, two thirds of it instruction fetch)
            const uint32_t* gbits = p.adj_bits + p.b_bits_off[b];
            const el_u32x4* rp = reinterpret_cast<const el_u32x4*>(gbits + (size_t)(rb * 32 + i) * (2 * HPW) + h * HPW);
            const el_u32x4 z4 = {0u, 0u, 0u, 0u};
            const el_u32x4 a0 = rp[0];
            const el_u32x4 a1 = HPW > 4 ? rp[1] : z4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { bitsw[j][tid] = a0[j]; bitsw[4 + j][tid] = a1[j]; }
        }
        const unsigned xbytes = (unsigned)(((size_t)(n - 1) * p.ldin + Fin) * 4);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.Hin) + (size_t)row0 * p.ldin, 0, (int)xbytes, 0x00020000);
        const int c = 32 * ct + i;
        const unsigned xvo = c < Fin ? (unsigned)((8 * h * p.ldin + c) * 4) : 0x80000000u;     // (a column past Fin reads zero: past any buffer, and no wrap with the row offset)
        const int xrow = p.ldin * 4;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        __syncthreads();                                          // the table
        auto request = [&](float (&d)[8], int s) {                // rows past n: offsets past the descriptor, zeros
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xvo, (16 * s + j) * xrow, 0));
        };
        auto multiply = [&](const float (&d)[8], int s) {
            el_bf16x8 a1, a2, a3;
            el_split8(d, a1, a2, a3);
            // the 8 bits of (row 32 rb + i, columns 16 s + 8 h ..): byte s & 3 of word s >> 2 of this lane's half row
            const unsigned pkw = bitsw[s >> 2][tid];
            const unsigned byte3 = ((pkw >> (8 * (s & 3))) & 0xFFu) << 3;
            const unsigned lo = byte3 & 0x78u, hi = (byte3 >> 4) & 0x78u;
            const el_u32x2 l2 = *reinterpret_cast<const el_u32x2*>(lut + lo);
            const el_u32x2 h2 = *reinterpret_cast<const el_u32x2*>(lut + hi);
            const el_u32x4 q = {l2.x, l2.y, h2.x, h2.y};
            const el_bf16x8 bq = __builtin_bit_cast(el_bf16x8, q);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bq, acc, 0, 0, 0);      // small planes first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bq, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq, acc, 0, 0, 0);
        };
        // this wave's steps s = kh + KS u, four to an iteration, three requests ahead of the one being multiplied
        float hb0[8], hb1[8], hb2[8], hb3[8];
        request(hb0, kh); request(hb1, kh + KS); request(hb2, kh + 2 * KS);
#pragma nounroll
        for (int s = kh; s < ksteps; s += 4 * KS) {               // wave-uniform
            request(hb3, s + 3 * KS);
            multiply(hb0, s);
            if (s + KS < ksteps) { request(hb0, s + 4 * KS); multiply(hb1, s + KS); }
            if (s + 2 * KS < ksteps) { request(hb1, s + 5 * KS); multiply(hb2, s + 2 * KS); }
            if (s + 3 * KS < ksteps) { request(hb2, s + 6 * KS); multiply(hb3, s + 3 * KS); }
        }
        // accumulator (r, lane): input column 32 ct + (r & 3) + 8 (r >> 2) + 4 h, output row i
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave][i][(r & 3) + 8 * (r >> 2) + 4 * h] = acc[r];
    }
    __syncthreads();
    {
        const int KP = (Fin + 15) & ~15;                          // the first Linear's contraction width (zero padded)
        const int KS = 4 / NCA;
        const float selfw = p.eps ? 1.f + p.eps[p.l] : 1.f;       // graphcnn.py:161 (1 + eps[layer]) h
        for (int c = c8; c < NCA * 32; c += 8) {
            float v = 0.f;
            for (int k = 0; k < KS; ++k) v += part[(c >> 5) + NCA * k][row][c & 31];
            const float hin = T1[row * kElTS + c];
            if (p.self_loop) v += hin;
            if (p.average) v /= deg;                              // 0 / 0 -> NaN as in the reference
            if (!p.self_loop) v += selfw * hin;
            if (c < KP) T0[row * kElTS + c] = (vrow && c < Fin) ? v : 0.f;
        }
    }
    // ---- B. the MLP --------------------------------------------------------------------------------------------
    float* Tin = T0;
    float* Tout = T1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k >= p.m) break;                                      // workgroup-uniform
        const int K = k == 0 ? Fin : H;
        const int ct = ctB, kh = khB, ncol = ncolB;
        const int nst = (K + 15) >> 4;
        __syncthreads();                                          // the input tile (and, the first time, the vectors) complete
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            auto stepB = [&](const float (&fb)[8], int s) {
                const int k0 = 16 * s + 8 * h;
                float fa[8];
                const float4 v0 = *reinterpret_cast<const float4*>(Tin + i * kElTS + k0);
                const float4 v1 = *reinterpret_cast<const float4*>(Tin + i * kElTS + k0 + 4);
                fa[0] = v0.x; fa[1] = v0.y; fa[2] = v0.z; fa[3] = v0.w; fa[4] = v1.x; fa[5] = v1.y; fa[6] = v1.z; fa[7] = v1.w;
                el_bf16x8 a1, a2, a3, b1, b2, b3;
                el_split8(fa, a1, a2, a3);
                el_split8(fb, b1, b2, b3);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);      // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            };
            // the wave's first two steps: W fragments requested at kernel entry; the rest (K = 128 with one k range per
            // wave) in a real loop, on demand
            if (kh < nst) stepB(fbw[k][0], kh);
            if (kh + KSB < nst) stepB(fbw[k][1], kh + KSB);
#pragma nounroll
            for (int s = kh + 2 * KSB; s < nst; s += KSB) {
                const int k0 = 16 * s + 8 * h;
                float fb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) fb[j] = k0 + j < K ? Wk[k][(size_t)ncol * ldwk[k] + k0 + j] : 0.f;
                stepB(fb, s);
            }
            // accumulator (r, lane): tile row (r & 3) + 8 (r >> 2) + 4 h, output column 32 ct + i
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = acc[r];
        }
        __syncthreads();
        const bool last = k == p.m - 1;
        for (int c = c8; c < H; c += 8) {
            float z = aff[k][0][c];
            for (int q = 0; q < KSB; ++q) z += part[(c >> 5) + NCT * q][row][c & 31];
            float y = gnm_relu(z * aff[k][1][c] + aff[k][2][c]);        // mlp.py:48 (inner) / graphcnn.py:163-166, 187-190 (outer)
            if (last) {
                if (vrow) p.Hout[(size_t)grow * p.ldh + c] = y;
                else y = 0.f;                                     // (rows past n: not part of the readout)
            }
            Tout[row * kElTS + c] = y;
        }
        __syncthreads();
        float* t = Tin; Tin = Tout; Tout = t;
    }
    // the block's share of the graph readout (graphcnn.py:228-229): column sums of its rows, fixed order
    if (tid < H) {
        float ssum = 0.f;
        for (int r = 0; r < 32; ++r) ssum += Tin[r * kElTS + tid];
        p.rpart[((size_t)b * p.wmax + rb) * H + tid] = ssum;
    }
}

struct ElFinArgs {
    const int32_t* node_off;
    const float* rpart;                           // [L][B][wmax][H]
    const long long* table;
    int B, wmax, L, m, H, C, graph_avg;
    float* g_f; int ldgf;
    float* c_sig;
    float* c_logit; int ldc;
};

__global__ void __launch_bounds__(256) gnm_eval_finish_kernel(const ElFinArgs p) {
    extern __shared__ float gfl[];                // [L * H]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = p.node_off[b + 1] - p.node_off[b];
    const int W = (n + 31) >> 5, H = p.H, LH = p.L * p.H;
    for (int e = tid; e < LH; e += 256) {
        const int l = e / H, c = e - l * H;
        float s = 0.f;
        for (int rb = 0; rb < W; ++rb) s += p.rpart[(((size_t)l * p.B + b) * p.wmax + rb) * H + c];
        if (p.graph_avg) s *= 1.0f / (float)n;    // the reference stores 1./len(graph.g) as fp32 (graphcnn.py:123,130)
        gfl[e] = s;
        p.g_f[(size_t)b * p.ldgf + e] = s;
        if (p.c_sig) p.c_sig[(size_t)b * p.ldgf + e] = 1.f / (1.f + expf(-s));
    }
    __syncthreads();
    // classifier head (graphcnn.py:224-231, eval: no dropout): a wave per class, lanes over the L*H products
    const int lane = tid & 63, wave = tid >> 6;
    for (int cls = wave; cls < p.C; cls += 4) {
        float acc = 0.f;
        for (int e = lane; e < LH; e += 64) {
            const int l = e / H, c = e - l * H;
            const long long* th = p.table + (size_t)p.L * p.m * kElLinWords + 2 * l;
            acc += gfl[e] * reinterpret_cast<const float*>(th[0])[(size_t)cls * H + c];
        }
        if (lane < p.L) acc += reinterpret_cast<const float*>((p.table + (size_t)p.L * p.m * kElLinWords + 2 * lane)[1])[cls];
        acc = wave_sum(acc);
        if (lane == 0) p.c_logit[(size_t)b * p.ldc + cls] = acc;
    }
}

// Floats of scratch gnm_eval_layers needs (the readout shares of every layer).
extern "C" long long gnm_eval_layers_scratch_floats(int B, int n_max, int H, int L) {
    return (long long)L * B * ((n_max + 31) / 32) * H;
}

// The eval-mode encoder + readout + classifier of B graphs as L + 1 launches (see the file header).  Arguments as
// gnm_eval_encoder (evalfwd.hip; the same DEVICE parameter table), with `scratch` (gnm_eval_layers_scratch_floats) in
// place of its two [N, H] arrays.  H in {32, 64, 128}, 1 <= m <= 3, F0 <= 128, C <= 256, every graph with a bit adjacency
// and at most 416 nodes: GNM_ERR_UNSUPPORTED otherwise (the caller then runs the layer-by-layer path).
extern "C" int gnm_eval_layers(const uint32_t* adj_bits, const int64_t* b_bits_off, const int32_t* node_off,
                               const int32_t* rowptr, const int64_t* b_rp_off, int B, int n_max, const float* X, int ldx,
                               int F0, int H, int L, int m, int C, int average, int self_loop, int graph_avg,
                               float bn_eps, const long long* table, const float* eps, float* hidden,
                               long long hidden_stride, int ldh, float* scratch, float* g_f, int ldgf, float* c_sig,
                               float* c_logit, int ldc, void* stream) {
    if (B <= 0) return GNM_OK;
    if (!(H == 32 || H == 64 || H == 128) || m < 1 || m > 3 || L < 1 || L > 16 || F0 < 1 || F0 > kElMaxH || C < 1 || C > 256 ||
        n_max < 1 || n_max > kElMaxN)
        return GNM_ERR_UNSUPPORTED;
    if (!adj_bits || !b_bits_off || !node_off || !rowptr || !b_rp_off || !X || !table || !hidden || !scratch || !g_f || !c_logit)
        return GNM_ERR_BAD_ARG;
    if (reinterpret_cast<uintptr_t>(adj_bits) & 15) return GNM_ERR_UNSUPPORTED;
    if ((long long)(n_max + 64) * (ldx > ldh ? ldx : ldh) * 4 >= (1LL << 31)) return GNM_ERR_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int wmax = (n_max + 31) / 32;
    for (int l = 0; l < L; ++l) {
        ElArgs a;
        memset(&a, 0, sizeof(a));
        a.adj_bits = adj_bits; a.b_bits_off = b_bits_off; a.node_off = node_off; a.rowptr = rowptr; a.b_rp_off = b_rp_off;
        a.Hin = l == 0 ? X : hidden + (size_t)(l - 1) * hidden_stride;
        a.ldin = l == 0 ? ldx : ldh;
        a.Fin = l == 0 ? F0 : H;
        a.B = B; a.wmax = wmax; a.L = L; a.m = m; a.l = l; a.H = H;
        a.average = average; a.self_loop = self_loop; a.bn_eps = bn_eps;
        a.eps = eps;
        a.table = table;
        a.Hout = hidden + (size_t)l * hidden_stride; a.ldh = ldh;
        a.rpart = scratch + (size_t)l * B * wmax * H;
        hipLaunchKernelGGL(gnm_eval_layer_kernel, dim3(B * wmax), dim3(256), 0, s, a);
        GNM_CHECK_LAUNCH();
    }
    ElFinArgs f;
    f.node_off = node_off; f.rpart = scratch; f.table = table; f.B = B; f.wmax = wmax; f.L = L; f.m = m; f.H = H; f.C = C;
    f.graph_avg = graph_avg; f.g_f = g_f; f.ldgf = ldgf; f.c_sig = c_sig; f.c_logit = c_logit; f.ldc = ldc;
    hipLaunchKernelGGL(gnm_eval_finish_kernel, dim3(B), dim3(256), (size_t)L * H * 4, s, f);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}
