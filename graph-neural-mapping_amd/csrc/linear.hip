// The per-layer MLP contraction of the GIN update (K4 of SURVEY.md section 2.2):
// nn.Linear at /root/reference models/mlp.py:25,32-35,43,48,49, forward and the two
// backward products autograd derives (dX = dZ W, dW = dZ^T X, db = sum dZ), on the
// exact-fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32; gfx950 has no xf32).
//
// Shapes are tall and skinny ([N, K] x [K, H], N ~ 4e5, K, H <= 128), so:
//   * the whole weight matrix lives in LDS, k-major, for the kernel's lifetime;
//   * each wave owns 32-row tiles of X.  A tile is staged through a wave-private,
//     padded LDS image with full-line 16-B loads (fragment-shaped global loads
//     would touch 32 lines per instruction), where the fused prologue
//     relu(x*scale+shift) -- the BatchNorm+ReLU of the previous Linear
//     (mlp.py:48) -- is applied on the fly;
//   * the MFMA k order is permuted (k = (KC/2)*half + step) so a lane's KC/2
//     A-operands are contiguous in LDS and come in with ds_read_b128, conflict-free
//     at a row stride of KC+4 floats;
//   * the epilogue adds the bias, stores Z, and accumulates per-column sum and sum
//     of squares (fp64) for the BatchNorm that follows (mlp.py:48, graphcnn.py:163):
//     in the 32x32 C/D layout a lane owns one output COLUMN, so the statistics are
//     register adds.
#include "gnm_common.h"

// tuning knobs, read once per process
static bool lin_force_generic() { static const bool v = getenv("GNM_LIN_GENERIC") != nullptr; return v; }
static bool lin_no_stream() { static const bool v = getenv("GNM_LIN_NO_STREAM") != nullptr; return v; }   // tuning: A/B
static bool linbwd_no_pipe() { static const bool v = getenv("GNM_LINBWD_NO_PIPE") != nullptr; return v; }   // tuning: A/B
static bool linbwd_no_samez() { static const bool v = getenv("GNM_LINBWD_NO_SAMEZ") != nullptr; return v; }

struct LinArgs {
    const float* X;
    const float* W;
    const float* bias;       // [H] or null
    float* Z;
    const float* pro_scale;  // [K] prologue x*scale+shift (then relu if pro_relu), or null
    const float* pro_shift;
    double* stats_partial;   // [gridDim.x][2][H] or null
    int ldx, ldw, ldz;
    int N, K, H;
    int w_kmajor;            // 0: W[h*ldw+k] (torch Linear weight);  1: W[k*ldw+h]
    int pro_relu;
    unsigned long long* stamps;   // tuning builds only (gnm_debug_set_lin_stamps): [block][4 waves][64] s_memtime
    int stat_rows;           // gnm_lin_split_kernel: rows of stats_partial = groups of four waves that take tiles
    int stage_out;           // gnm_lin_kernel: the launch reserved LDS for the output staging image
    // gnm_lin_split128_kernel, dX form (gnm_linear_dgrad_masked): the output is the gradient arriving at relu(bn(mZ)) of
    // the BatchNorm + ReLU below -- apply that ReLU mask on the way out and reduce that BatchNorm's backward sums
    // (sum G, sum G xhat) into stats_partial instead of the forward's (sum Z, sum Z^2)
    const float* mZ; const float* m_scale; const float* m_shift; const float* m_mean; const float* m_rstd;
    int ldmz;
};

// First tile of a wave in the strided tile walk (stride = all active waves of the launch).  Numbering the waves
// workgroup by workgroup hands the remainder tiles (ntiles mod stride) to the FIRST workgroups only -- at the headline
// shape 12,800 tiles over 3,072 waves: five tiles for every wave of workgroups 0-42 and four for the rest, and the
// launch lasts as long as those 43 CUs need for 60 tiles while the other 213 have 48.  Numbered slot-major instead
// (one wave of every group of every workgroup first, on different SIMDs, then the next ...) every CU gets 50.
// `groups` groups of four waves per workgroup, `rows` groups in the launch; a launch whose last workgroup is not full
// (rows != groups * gridDim.x: small N) keeps the plain numbering, and waves past the last row get no tile.
__host__ __device__ __forceinline__ int lin_first_tile_of(int wave, int groups, int rows, int nwg, int b) {
    const int grp = wave >> 2;
    if (rows == groups * nwg) {
        const int slot = ((wave & 3) + grp * (groups == 2 ? 2 : 1)) & 3;
        return (slot * groups + grp) * nwg + b;
    }
    const int gw = b * groups * 4 + wave;
    return gw < 4 * rows ? gw : 0x3fffffff;
}
__device__ __forceinline__ int lin_first_tile(int wave, int groups, int rows) {
    return lin_first_tile_of(wave, groups, rows, (int)gridDim.x, (int)blockIdx.x);
}
// The same map on the host, for tests/test_host_logic.py: every tile index below 4 x rows must belong to exactly one
// (workgroup, wave) of the launch, whatever the launch shape.
extern "C" int gnm_debug_lin_first_tile(int wave, int groups, int rows, int nwg, int b) {
    return lin_first_tile_of(wave, groups, rows, nwg, b);
}

#ifdef GNM_LIN_TUNING
static unsigned long long* g_lin_stamps = nullptr;
extern "C" void gnm_debug_set_lin_stamps(void* p) { g_lin_stamps = reinterpret_cast<unsigned long long*>(p); }
#define GNM_LSTAMP(k)                                                                                         \
    if (p.stamps && (threadIdx.x & 63) == 0)                                                                  \
        p.stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + (k)] = __builtin_amdgcn_s_memtime();
#define GNM_RSTAMP(k)      /* gnm_linear_bwd_rz_kernel: eight waves per workgroup */                        \
    if (p.stamps && (threadIdx.x & 63) == 0)                                                                  \
        p.stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + (k)] = __builtin_amdgcn_s_memtime();
#define GNM_SSTAMP(k)      /* gnm_lin_split_kernel: twelve waves per workgroup, six stamps per tile */        \
    if (p.stamps && (threadIdx.x & 63) == 0)                                                                  \
        p.stamps[((size_t)blockIdx.x * kSplitWaves + (threadIdx.x >> 6)) * 64 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define GNM_LSTAMP(k)
#define GNM_RSTAMP(k)
#define GNM_SSTAMP(k)
#endif

template <int KC, int HT>
__global__ void __launch_bounds__(256) gnm_lin_kernel(const LinArgs p) {
    constexpr int HP = HT * 32;          // padded output width
    constexpr int XS = KC + 4;           // staging row stride (floats): conflict-free ds_read_b128
    constexpr int C4 = KC / 4;           // float4 per staged row
    constexpr int KH = KC / 2;           // MFMA steps per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nchunks = (p.K + KC - 1) / KC;
    const int KP = nchunks * KC;
    float* Wt = reinterpret_cast<float*>(smem);                   // [KP][HP]
    float* Xs_all = Wt + (size_t)KP * HP;                         // [4][32][XS]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    const int h = lane >> 5;
    float* Xs = Xs_all + wave * 32 * XS;
    // output staging (aligned full-width outputs): the tile leaves as 16-byte row-contiguous stores -- 1 KiB per
    // wave-instruction instead of 128-byte column pieces; the input layer's Linear (K = F0) is all output traffic
    constexpr int OS = HP + 4, O4 = HP / 4;
    float* Os = Xs_all + 4 * 32 * XS + wave * 32 * OS;
    const bool vec_out = p.stage_out && p.H == HP && (p.ldz & 3) == 0 && (reinterpret_cast<uintptr_t>(p.Z) & 15) == 0;

    // ---- stage W^T (k-major, zero padded) --------------------------------------
    if (p.w_kmajor) {
        for (int idx = tid; idx < KP * HP; idx += 256) {
            const int k = idx / HP, hh = idx - k * HP;
            Wt[idx] = (k < p.K && hh < p.H) ? p.W[(size_t)k * p.ldw + hh] : 0.f;
        }
    } else {
        for (int idx = tid; idx < KP * HP; idx += 256) {
            const int hh = idx / KP, k = idx - hh * KP;
            Wt[k * HP + hh] = (k < p.K && hh < p.H) ? p.W[(size_t)hh * p.ldw + k] : 0.f;
        }
    }
    __syncthreads();

    const bool vec_in = ((p.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.X) & 15) == 0);
    const int ntiles = (p.N + 31) / 32;
    double st1[HT], st2[HT];
#pragma unroll
    for (int c = 0; c < HT; ++c) { st1[c] = 0.0; st2[c] = 0.0; }
    float bias_r[HT];
#pragma unroll
    for (int c = 0; c < HT; ++c) {
        const int col = 32 * c + i;
        bias_r[c] = (p.bias && col < p.H) ? p.bias[col] : 0.f;
    }
    const int c4 = lane % C4;            // loop-invariant: 64 % C4 == 0

    for (int t = lin_first_tile(wave, 1, gridDim.x); t < ntiles; t += gridDim.x * 4) {
        const int r0 = t * 32;
        f32x16 acc[HT];
#pragma unroll
        for (int c = 0; c < HT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

        for (int kc = 0; kc < nchunks; ++kc) {
            const int kb = kc * KC;
            // -- stage the [32][KC] chunk of X (coalesced), fused prologue --
            const int k4 = kb + 4 * c4;
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.pro_scale) {
                sc.x = (k4 + 0 < p.K) ? p.pro_scale[k4 + 0] : 0.f; sh.x = (k4 + 0 < p.K) ? p.pro_shift[k4 + 0] : 0.f;
                sc.y = (k4 + 1 < p.K) ? p.pro_scale[k4 + 1] : 0.f; sh.y = (k4 + 1 < p.K) ? p.pro_shift[k4 + 1] : 0.f;
                sc.z = (k4 + 2 < p.K) ? p.pro_scale[k4 + 2] : 0.f; sh.z = (k4 + 2 < p.K) ? p.pro_shift[k4 + 2] : 0.f;
                sc.w = (k4 + 3 < p.K) ? p.pro_scale[k4 + 3] : 0.f; sh.w = (k4 + 3 < p.K) ? p.pro_shift[k4 + 3] : 0.f;
            }
#pragma unroll
            for (int idx = lane; idx < 32 * C4; idx += 64) {
                const int row = idx / C4;
                const int grow = r0 + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (grow < p.N) {
                    const float* src = p.X + (size_t)grow * p.ldx + k4;
                    if (vec_in && k4 + 3 < p.K) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (k4 + 0 < p.K) v.x = src[0];
                        if (k4 + 1 < p.K) v.y = src[1];
                        if (k4 + 2 < p.K) v.z = src[2];
                        if (k4 + 3 < p.K) v.w = src[3];
                    }
                    if (p.pro_scale) {
                        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y;
                        v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                        if (p.pro_relu) {
                            v.x = gnm_relu(v.x); v.y = gnm_relu(v.y);
                            v.z = gnm_relu(v.z); v.w = gnm_relu(v.w);
                        }
                    }
                }
                *reinterpret_cast<float4*>(Xs + row * XS + 4 * c4) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // -- A fragments: lane (i, h) takes X[r0+i][kb + KH*h + s], s = 0..KH-1 --
            float a[KH];
#pragma unroll
            for (int j = 0; j < KH / 4; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 4 * j);
                a[4 * j + 0] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
            }
            const float* wrow = Wt + (size_t)(kb + KH * h) * HP + i;
#pragma unroll
            for (int s = 0; s < KH; ++s) {
#pragma unroll
                for (int c = 0; c < HT; ++c) {
                    const float bv = wrow[s * HP + 32 * c];
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv, acc[c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // A fragments are in registers before Xs is overwritten
        }

        // ---- epilogue: bias, store, BatchNorm statistics ----------------------
#pragma unroll
        for (int c = 0; c < HT; ++c) {
            const int col = 32 * c + i;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int grow = r0 + lrow;
                const float z = acc[c][r] + bias_r[c];
                if (vec_out) Os[lrow * OS + col] = z;
                if (grow < p.N && col < p.H) {
                    if (!vec_out) p.Z[(size_t)grow * p.ldz + col] = z;
                    s1 += z;
                    s2 += z * z;
                }
            }
            st1[c] += (double)s1;
            st2[c] += (double)s2;
        }
        if (vec_out) {                                           // (kernel-uniform)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int idx = lane; idx < 32 * O4; idx += 64) {
                const int row = idx / O4, oc = idx - row * O4;
                if (r0 + row < p.N)
                    *reinterpret_cast<float4*>(p.Z + (size_t)(r0 + row) * p.ldz + 4 * oc) =
                        *reinterpret_cast<const float4*>(Os + row * OS + 4 * oc);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

    if (p.stats_partial) {   // block-level combine in a fixed order, one partial row per block
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);   // [4 waves][2][HP]
#pragma unroll
        for (int c = 0; c < HT; ++c) {
            const double a1 = st1[c] + __shfl_xor(st1[c], 32, 64);
            const double a2 = st2[c] + __shfl_xor(st2[c], 32, 64);
            if (h == 0) {
                red[(wave * 2 + 0) * HP + 32 * c + i] = a1;
                red[(wave * 2 + 1) * HP + 32 * c + i] = a2;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * HP; idx += 256) {
            const int which = idx / HP, col = idx - which * HP;
            if (col < p.H) {
                double s = 0.0;
                for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * HP + col];
                p.stats_partial[((size_t)blockIdx.x * 2 + which) * p.H + col] = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Pipelined variant for the common case: 16-B aligned rows, K a multiple of the chunk KC,
// H == 32*HT.  No column guards; rows past N are clamped on load and masked in the
// epilogue.  Differences from the generic kernel above:
//   * the NEXT tile's global loads are issued right after this tile's A fragments are in
//     registers, so they fly under the MFMAs (the generic kernel serialises load -> MFMA);
//   * the output tile is transposed through the wave's staging image and stored as full
//     16-B row-contiguous chunks (1 KiB per wave-instruction) instead of 4-B column pieces.
// ---------------------------------------------------------------------------------
template <int KC, int HT>
__global__ void __launch_bounds__(256) gnm_lin_fast_kernel(const LinArgs p) {
    constexpr int HP = HT * 32;
    constexpr int XS = (KC > HP ? KC : HP) + 4;   // staging row stride: holds an input chunk or the output tile
    constexpr int C4 = KC / 4;
    constexpr int KH = KC / 2;
    constexpr int NLD = (32 * C4) / 64;           // float4 loads per lane per chunk (KC >= 8)
    constexpr int O4 = HP / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nchunks = p.K / KC;
    float* Wt = reinterpret_cast<float*>(smem);                   // [K][HP]
    float* Xs_all = Wt + (size_t)p.K * HP;                        // [4][32][XS]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    const int h = lane >> 5;
    float* Xs = Xs_all + wave * 32 * XS;
    GNM_LSTAMP(0)

    if (p.w_kmajor) {
        for (int idx = tid; idx < p.K * HP; idx += 256) {
            const int k = idx / HP, hh = idx - k * HP;
            Wt[idx] = p.W[(size_t)k * p.ldw + hh];
        }
    } else {
        for (int idx = tid; idx < p.K * HP; idx += 256) {
            const int hh = idx / p.K, k = idx - hh * p.K;
            Wt[k * HP + hh] = p.W[(size_t)hh * p.ldw + k];
        }
    }
    __syncthreads();
    GNM_LSTAMP(1)
    int tk = 0;

    const int ntiles = (p.N + 31) / 32;
    double st1[HT], st2[HT];
    float bias_r[HT];
#pragma unroll
    for (int c = 0; c < HT; ++c) {
        st1[c] = 0.0; st2[c] = 0.0;
        bias_r[c] = p.bias ? p.bias[32 * c + i] : 0.f;
    }
    const int c4 = lane % C4;
    const int lrow0 = lane / C4;                  // row of this lane's first float4; later ones are +64/C4 rows
    constexpr int RSTEP = 64 / C4;
    const int tstride = gridDim.x * 4;
    const int nlast = p.N - 1;

    float4 raw[NLD];
    int t = lin_first_tile(wave, 1, gridDim.x);
    if (t < ntiles) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int grow = min(t * 32 + lrow0 + j * RSTEP, nlast);
            raw[j] = *reinterpret_cast<const float4*>(p.X + (size_t)grow * p.ldx + 4 * c4);
        }
    }
    for (; t < ntiles; t += tstride) {
        const int r0 = t * 32;
        GNM_LSTAMP(2 + 4 * min(tk, 14))
        f32x16 acc[HT];
#pragma unroll
        for (int c = 0; c < HT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

        for (int kc = 0; kc < nchunks; ++kc) {
            const int kb = kc * KC;
            if (p.pro_scale) {
                const float4 sc = *reinterpret_cast<const float4*>(p.pro_scale + kb + 4 * c4);
                const float4 sh = *reinterpret_cast<const float4*>(p.pro_shift + kb + 4 * c4);
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    float4 v = raw[j];
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    if (p.pro_relu) {
                        v.x = gnm_relu(v.x); v.y = gnm_relu(v.y); v.z = gnm_relu(v.z); v.w = gnm_relu(v.w);
                    }
                    raw[j] = v;
                }
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j)
                *reinterpret_cast<float4*>(Xs + (lrow0 + j * RSTEP) * XS + 4 * c4) = raw[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float a[KH];
#pragma unroll
            for (int j = 0; j < KH / 4; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 4 * j);
                a[4 * j + 0] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
            }
            // next chunk (of this tile, or chunk 0 of this wave's next tile): loads fly under the MFMAs
            {
                const bool more_k = kc + 1 < nchunks;
                const int tn = more_k ? t : t + tstride;
                const int kn = more_k ? kb + KC : 0;
                if (tn < ntiles) {
#pragma unroll
                    for (int j = 0; j < NLD; ++j) {
                        const int grow = min(tn * 32 + lrow0 + j * RSTEP, nlast);
                        raw[j] = *reinterpret_cast<const float4*>(p.X + (size_t)grow * p.ldx + kn + 4 * c4);
                    }
                }
            }
            GNM_LSTAMP(3 + 4 * min(tk, 14))
            const float* wrow = Wt + (size_t)(kb + KH * h) * HP + i;
#pragma unroll
            for (int s = 0; s < KH; ++s) {
#pragma unroll
                for (int c = 0; c < HT; ++c) {
                    const float bv = wrow[s * HP + 32 * c];
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv, acc[c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        GNM_LSTAMP(4 + 4 * min(tk, 14))

        // epilogue: bias, statistics (valid rows only), transpose through LDS, 16-B stores
#pragma unroll
        for (int c = 0; c < HT; ++c) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float z = acc[c][r] + bias_r[c];
                Xs[lrow * XS + 32 * c + i] = z;
                if (r0 + lrow < p.N) {
                    s1 += z;
                    s2 += z * z;
                }
            }
            st1[c] += (double)s1;
            st2[c] += (double)s2;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int idx = lane; idx < 32 * O4; idx += 64) {
            const int row = idx / O4, oc = idx - row * O4;
            if (r0 + row < p.N)
                *reinterpret_cast<float4*>(p.Z + (size_t)(r0 + row) * p.ldz + 4 * oc) =
                    *reinterpret_cast<const float4*>(Xs + row * XS + 4 * oc);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GNM_LSTAMP(5 + 4 * min(tk, 14))
        ++tk;
    }
    GNM_LSTAMP(62)

    if (p.stats_partial) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);   // [4 waves][2][HP]
#pragma unroll
        for (int c = 0; c < HT; ++c) {
            const double a1 = st1[c] + __shfl_xor(st1[c], 32, 64);
            const double a2 = st2[c] + __shfl_xor(st2[c], 32, 64);
            if (h == 0) {
                red[(wave * 2 + 0) * HP + 32 * c + i] = a1;
                red[(wave * 2 + 1) * HP + 32 * c + i] = a2;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * HP; idx += 256) {
            const int which = idx / HP, col = idx - which * HP;
            double s = 0.0;
            for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * HP + col];
            p.stats_partial[((size_t)blockIdx.x * 2 + which) * p.H + col] = s;
        }
    }
}

template <int KC, int HT>
static int launch_lin_fast(const LinArgs& a, int grid, hipStream_t s) {
    constexpr int HP = HT * 32;
    constexpr int XS = (KC > HP ? KC : HP) + 4;
    size_t lds = (size_t)a.K * HP * 4 + (size_t)4 * 32 * XS * 4;
    const size_t red = (size_t)4 * 2 * HP * 8;
    if (red > lds) lds = red;
    if (lds > (size_t)kLdsBudget) return GNM_ERR_UNSUPPORTED;
    GNM_ALLOW_FULL_LDS((&gnm_lin_fast_kernel<KC, HT>));
    hipLaunchKernelGGL((gnm_lin_fast_kernel<KC, HT>), dim3(grid), dim3(256), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------
// Streaming variant for K == KC (the whole input row is one chunk: every hidden Linear of the model).  Same tile
// arithmetic as the pipelined kernel above; what changes is how a wave's memory operations are ordered and counted.
// gfx950 retires vector-memory operations IN ORDER, stores included, and `s_waitcnt vmcnt(n)` waits until at most n are
// outstanding.  The pipelined kernel issues  loads(t+1) .. stores(t)  and then needs loads(t+1): the compiler could
// not prove that the stores behind them had been issued on every path (row guards, the first trip through the loop),
// so it waited for vmcnt(7..0) -- i.e. for the STORES of the previous tile to land in HBM -- before staging the next
// tile, and for vmcnt(0) (bias load, never counted down) in front of every epilogue.  The in-kernel timeline showed
// those two waits as ~22 % + part of the epilogue's 25 % of a tile's time.  Here
//   * loads and stores are buffer instructions whose descriptor covers exactly the tile's valid rows: rows past N
//     (and whole tiles past the last) are dropped by the address check, so every access is issued unconditionally;
//   * the first tile is peeled, so the loop is entered with the same  loads, stores  queue its back edge carries;
//   * bias / prologue constants are waited for once, before the first tile.
// The wait in front of the staging writes becomes vmcnt(15..8): the previous tile's stores drain under this tile's
// MFMAs instead of in front of them.
// ---------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gnm_tile_rsrc(const float* base, long long rows, int ld, int width) {
    // raw buffer (stride 0): byte offsets >= num_records read 0 / are not written.  On gfx950 the check covers the
    // scalar offset too (tools/ubench/soffset_check.hip: a load that leaves the range only through soffset returns 0),
    // so the loads below may carry their row step in soffset; the STORES carry it in the vector offset for another
    // reason (the data-register hazard described at gnm_lin_stream_kernel).
#ifdef GNM_ABLATE_TILE_TRAFFIC       // tuning builds: every tile descriptor empty -- the kernels' time without tile traffic
    const unsigned bytes = 0u;
    (void)rows; (void)ld; (void)width;
#else
    const unsigned bytes = rows > 0 ? (unsigned)(((rows - 1) * ld + width) * 4) : 0u;
#endif
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}

template <int KC, int HT>
__global__ void __launch_bounds__(256) gnm_lin_stream_kernel(const LinArgs p) {
    constexpr int HP = HT * 32;
    constexpr int XS = (KC > HP ? KC : HP) + 4;
    constexpr int C4 = KC / 4;
    constexpr int KH = KC / 2;
    constexpr int NLD = (32 * C4) / 64;           // 16-B loads per lane per tile
    constexpr int RSTEP = 64 / C4;                // rows between a lane's consecutive loads
    constexpr int O4 = HP / 4;
    constexpr int NST = (32 * O4) / 64;           // 16-B stores per lane per tile
    constexpr int WSTEP = 64 / O4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Wt = reinterpret_cast<float*>(smem);                   // [KC][HP]
    float* Xs_all = Wt + (size_t)KC * HP;                         // [4][32][XS]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    const int h = lane >> 5;
    float* Xs = Xs_all + wave * 32 * XS;
    GNM_LSTAMP(0)

    const int c4 = lane % C4;
    const int lrow0 = lane / C4;
    float bias_r[HT];
#pragma unroll
    for (int c = 0; c < HT; ++c) bias_r[c] = p.bias ? p.bias[32 * c + i] : 0.f;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool pro = p.pro_scale != nullptr;
    if (pro) {
        sc = *reinterpret_cast<const float4*>(p.pro_scale + 4 * c4);
        sh = *reinterpret_cast<const float4*>(p.pro_shift + 4 * c4);
    }
    const int ntiles = (p.N + 31) / 32;
    const int tstride = gridDim.x * 4;
    const int in_voff = (lrow0 * p.ldx + 4 * c4) * 4;            // byte offsets inside a tile's descriptor
    const int in_step = RSTEP * p.ldx * 4;
    const int out_voff = ((lane / O4) * p.ldz + 4 * (lane % O4)) * 4;
    const int out_step = WSTEP * p.ldz * 4;

    u32x4 raw[NLD];
    auto load_tile = [&](int tile) {
        const long long row0 = (long long)tile * 32;
        const __amdgpu_buffer_rsrc_t rs = gnm_tile_rsrc(p.X + row0 * p.ldx, min((long long)p.N - row0, 32LL), p.ldx, KC);
#pragma unroll
        for (int j = 0; j < NLD; ++j) raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, in_voff, j * in_step, 0);
    };
    int t = lin_first_tile(wave, 1, gridDim.x);
    load_tile(t);                                 // the first tile arrives while the weight is staged

    // weight -> LDS as Wt[k][h].  torch layout W[h][k]: a lane takes 4 consecutive k of one h (16-B global read) and
    // consecutive lanes take consecutive h, so the four transposed LDS writes of a wave fall in consecutive banks.
    // (a weight inside a flat parameter buffer need not be 16-byte aligned: four 4-byte loads then)
    const bool w_vec = (p.ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0;
    auto load_w4 = [&](const float* src) -> float4 {
        if (w_vec) return *reinterpret_cast<const float4*>(src);
        return make_float4(src[0], src[1], src[2], src[3]);
    };
    if (p.w_kmajor) {
#pragma unroll
        for (int it = 0; it < (KC * O4) / 256; ++it) {
            const int idx = tid + 256 * it;
            const int k = idx / O4, h4 = idx - k * O4;
            *reinterpret_cast<float4*>(Wt + k * HP + 4 * h4) = load_w4(p.W + (size_t)k * p.ldw + 4 * h4);
        }
    } else {
        float4 w[(C4 * HP) / 256];
#pragma unroll
        for (int it = 0; it < (C4 * HP) / 256; ++it) {
            const int idx = tid + 256 * it;
            const int k4 = idx / HP, hh = idx - k4 * HP;
            w[it] = load_w4(p.W + (size_t)hh * p.ldw + 4 * k4);
        }
#pragma unroll
        for (int it = 0; it < (C4 * HP) / 256; ++it) {
            const int idx = tid + 256 * it;
            const int k4 = idx / HP, hh = idx - k4 * HP;
            Wt[(4 * k4 + 0) * HP + hh] = w[it].x;
            Wt[(4 * k4 + 1) * HP + hh] = w[it].y;
            Wt[(4 * k4 + 2) * HP + hh] = w[it].z;
            Wt[(4 * k4 + 3) * HP + hh] = w[it].w;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): nothing from the preamble is pending inside the tile loop
    __syncthreads();
    GNM_LSTAMP(1)
    int tk = 0;

    double st1[HT], st2[HT];
#pragma unroll
    for (int c = 0; c < HT; ++c) { st1[c] = 0.0; st2[c] = 0.0; }
    auto do_tile = [&](int t) {
        const int r0 = t * 32;
        GNM_LSTAMP(2 + 4 * min(tk, 14))
        f32x16 acc[HT];
#pragma unroll
        for (int c = 0; c < HT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            float4 v = __builtin_bit_cast(float4, raw[j]);
            if (pro) {
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (p.pro_relu) {
                    v.x = gnm_relu(v.x); v.y = gnm_relu(v.y); v.z = gnm_relu(v.z); v.w = gnm_relu(v.w);
                }
            }
            *reinterpret_cast<float4*>(Xs + (lrow0 + j * RSTEP) * XS + 4 * c4) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float a[KH];
#pragma unroll
        for (int j = 0; j < KH / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 4 * j);
            a[4 * j + 0] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
        }
        load_tile(t + tstride);                   // past the last tile: an empty descriptor, no memory traffic
        GNM_LSTAMP(3 + 4 * min(tk, 14))
        const float* wrow = Wt + (size_t)(KH * h) * HP + i;
#pragma unroll
        for (int s = 0; s < KH; ++s) {
#pragma unroll
            for (int c = 0; c < HT; ++c) {
                const float bv = wrow[s * HP + 32 * c];
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bv, acc[c], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // A fragments are in registers before the staging image is reused
        GNM_LSTAMP(4 + 4 * min(tk, 14))
#pragma unroll
        for (int c = 0; c < HT; ++c) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float z = acc[c][r] + bias_r[c];
                Xs[lrow * XS + 32 * c + i] = z;
                if (r0 + lrow < p.N) {
                    s1 += z;
                    s2 += z * z;
                }
            }
            st1[c] += (double)s1;
            st2[c] += (double)s2;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const __amdgpu_buffer_rsrc_t rz = gnm_tile_rsrc(p.Z + (size_t)r0 * p.ldz, min(p.N - r0, 32), p.ldz, HP);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int idx = lane + 64 * st;
            const int row = idx / O4, oc = idx - row * O4;
            const u32x4 v = *reinterpret_cast<const u32x4*>(Xs + row * XS + 4 * oc);
            // row step in the VECTOR offset, scalar offset 0: with an SGPR soffset hipcc (ROCm 7.2) schedules a VALU
            // write of the data registers straight behind a 16-byte buffer store (its hazard table exempts that form)
            // and on gfx950 the store then picks up the new value in some lanes -- seen as address integers in Z.
            __builtin_amdgcn_raw_buffer_store_b128(v, rz, out_voff + st * out_step, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GNM_LSTAMP(5 + 4 * min(tk, 14))
        ++tk;
    };

    if (t < ntiles) {
        do_tile(t);                               // peeled: the loop below starts with loads(t+1), stores(t) in flight
        for (t += tstride; t < ntiles; t += tstride) do_tile(t);
    }
    GNM_LSTAMP(62)

    if (p.stats_partial) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);   // [4 waves][2][HP]
#pragma unroll
        for (int c = 0; c < HT; ++c) {
            const double a1 = st1[c] + __shfl_xor(st1[c], 32, 64);
            const double a2 = st2[c] + __shfl_xor(st2[c], 32, 64);
            if (h == 0) {
                red[(wave * 2 + 0) * HP + 32 * c + i] = a1;
                red[(wave * 2 + 1) * HP + 32 * c + i] = a2;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * HP; idx += 256) {
            const int which = idx / HP, col = idx - which * HP;
            double s = 0.0;
            for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * HP + col];
            p.stats_partial[((size_t)blockIdx.x * 2 + which) * p.H + col] = s;
        }
    }
}

template <int KC, int HT>
static int launch_lin_stream(const LinArgs& a, int grid, hipStream_t s) {
    constexpr int HP = HT * 32;
    constexpr int XS = (KC > HP ? KC : HP) + 4;
    size_t lds = (size_t)KC * HP * 4 + (size_t)4 * 32 * XS * 4;
    const size_t red = (size_t)4 * 2 * HP * 8;
    if (red > lds) lds = red;
    if (lds > (size_t)kLdsBudget) return GNM_ERR_UNSUPPORTED;
    GNM_ALLOW_FULL_LDS((&gnm_lin_stream_kernel<KC, HT>));
    hipLaunchKernelGGL((gnm_lin_stream_kernel<KC, HT>), dim3(grid), dim3(256), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------
// Split-precision variant of the streaming kernel for K = 64, H = 64 (every hidden Linear of the headline model).
// The fp32 matrix instruction (v_mfma_f32_32x32x2_f32, 64 FLOP/clk per SIMD) kept a wave in its product for 41 % of
// its life with three waves per SIMD -- the matrix pipe was the resource the waves queued for, not HBM.  Here both
// operands are split by truncation into three bf16 planes each (x = x1 + x2 + x3 EXACTLY: 8 + 8 + 8 mantissa bits) and
// the product runs as six bf16 matrix instructions per 16 k (x1w1, x1w2, x2w1, x2w2, x1w3, x3w1; 16x the fp32 rate):
// every partial product is exact in the fp32 accumulator, the three dropped terms are below 2^-24 of |x||w| each --
// the size of the accumulator's own rounding -- and the instruction count falls from 64 x 64 to 48 x 32 cycles per tile.
// One 768-thread workgroup per CU (12 waves share one 24 KB image of the weight planes; three 256-thread workgroups
// with a copy each do not fit the LDS); its waves form three groups of four that write three rows of statistics
// partials, so the launch produces exactly the gnm_linear_grid(N) rows the BatchNorm finalize expects.
// ---------------------------------------------------------------------------------
typedef __bf16 lin_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void lin_split3(const float f, unsigned& a1, unsigned& a2, unsigned& a3) {
    a1 = __float_as_uint(f) & 0xFFFF0000u;
    const float r1 = f - __uint_as_float(a1);
    a2 = __float_as_uint(r1) & 0xFFFF0000u;
    a3 = __float_as_uint(r1 - __uint_as_float(a2));        // <= 8 significant bits left: its top half is all of it
}
// (top 16 bits of hi_word) : (top 16 bits of lo_word)
__device__ __forceinline__ unsigned lin_bf16_pair(unsigned lo_word, unsigned hi_word) {
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);
}
// eight consecutive-k floats -> the three bf16x8 operands
__device__ __forceinline__ void lin_split8(const float* f, u32x4& p1, u32x4& p2, u32x4& p3) {
    unsigned a1[8], a2[8], a3[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) lin_split3(f[j], a1[j], a2[j], a3[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p1[j] = lin_bf16_pair(a1[2 * j], a1[2 * j + 1]);
        p2[j] = lin_bf16_pair(a2[2 * j], a2[2 * j + 1]);
        p3[j] = lin_bf16_pair(a3[2 * j], a3[2 * j + 1]);
    }
}

#ifndef GNM_SPLIT_WAVES           // tuning builds: 16 = four waves per SIMD (needs GNM_LIN_GRID=1024 to fill 256 CUs)
#define GNM_SPLIT_WAVES 12
#endif
static constexpr int kSplitWaves = GNM_SPLIT_WAVES;
#ifndef GNM_L64_ABLATE           // tuning builds (-DGNM_L64_ABLATE=n): 1 = no output stores, 2 = no input traffic
#define GNM_L64_ABLATE 0
#endif

template <int HT>
__global__ void __launch_bounds__(kSplitWaves * 64) gnm_lin_split_kernel(const LinArgs p) {
    constexpr int KC = 64, NW = kSplitWaves, NT = NW * 64;
    constexpr int HP = HT * 32;
    constexpr int XS = (KC > HP ? KC : HP) + 4;
    constexpr int C4 = KC / 4;
    constexpr int KH = KC / 2;
    constexpr int NLD = (32 * C4) / 64;
    constexpr int RSTEP = 64 / C4;
    constexpr int O4 = HP / 4;
    constexpr int NST = (32 * O4) / 64;
    constexpr int WSTEP = 64 / O4;
    constexpr int E = 4 * HT * 64;                // operand entries (16 B) per weight plane: [m][c][lane]
    GNM_SSTAMP(0)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* Wp = reinterpret_cast<u32x4*>(smem);                               // [3][E]
    float* Xs_all = reinterpret_cast<float*>(smem + (size_t)3 * E * 16);      // [NW][32][XS]; first the fp32 weight image
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    const int h = lane >> 5;
    float* Xs = Xs_all + wave * 32 * XS;

    const int c4 = lane % C4;
    const int lrow0 = lane / C4;
    float bias_r[HT];
#pragma unroll
    for (int c = 0; c < HT; ++c) bias_r[c] = p.bias ? p.bias[32 * c + i] : 0.f;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool pro = p.pro_scale != nullptr;
    if (pro) {
        sc = *reinterpret_cast<const float4*>(p.pro_scale + 4 * c4);
        sh = *reinterpret_cast<const float4*>(p.pro_shift + 4 * c4);
    }
    // waves are numbered across the launch; groups of four own one row of statistics partials, exactly the tile ->
    // row map of the 4-wave kernels; waves past the last row (a grid that is not a multiple of 3) take no tiles
    const int gw = lin_first_tile(wave, NW / 4, p.stat_rows);
    const int ntiles = (p.N + 31) / 32;
    const int tstride = 4 * p.stat_rows;
    const int in_voff = (lrow0 * p.ldx + 4 * c4) * 4;
    const int in_step = RSTEP * p.ldx * 4;
    const int out_voff = ((lane / O4) * p.ldz + 4 * (lane % O4)) * 4;
    const int out_step = WSTEP * p.ldz * 4;

    u32x4 raw[NLD];
    auto load_tile = [&](int tile) {
        const long long row0 = (long long)tile * 32;
        const __amdgpu_buffer_rsrc_t rs = gnm_tile_rsrc(p.X + row0 * p.ldx, (GNM_L64_ABLATE & 2) ? 0LL : min((long long)p.N - row0, 32LL), p.ldx, KC);
#pragma unroll
        for (int j = 0; j < NLD; ++j) raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, in_voff, j * in_step, 0);
    };
    int t = gw;
    load_tile(t);

    // weight -> fp32 image Wt[k][h] in the staging region, then -> the three bf16 operand planes
    float* Wt = Xs_all;
    const bool w_vec = (p.ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0;
    auto load_w4 = [&](const float* src) -> float4 {
        if (w_vec) return *reinterpret_cast<const float4*>(src);
        return make_float4(src[0], src[1], src[2], src[3]);
    };
    // entry (m, c, lane = 32 kg + n): W[k = 8 m + 32 kg + 0..7][h = 32 c + n] -- the k numbering the A fragments use
    if (!p.w_kmajor) {
        // torch's layout W[h][k]: an entry's eight k are 32 contiguous bytes of row h -- straight from global memory
        // (L2) into the split; the fp32 image, its transposing scalar LDS writes and one of the two barriers were a
        // fifth of a wave's lifetime (in-kernel timeline, tools/lin_timeline.py --kernel fwd_split)
        for (int e = tid; e < E; e += NT) {
            const int n = e & 31, kg = (e >> 5) & 1, c = (e >> 6) % HT, m = e / (64 * HT);
            const float* src = p.W + (size_t)(32 * c + n) * p.ldw + 8 * m + 32 * kg;
            const float4 w0 = load_w4(src), w1 = load_w4(src + 4);
            const float f[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            u32x4 p1, p2, p3;
            lin_split8(f, p1, p2, p3);
            Wp[e] = p1; Wp[E + e] = p2; Wp[2 * E + e] = p3;
        }
    } else {
        for (int idx = tid; idx < KC * O4; idx += NT) {
            const int k = idx / O4, h4 = idx - k * O4;
            *reinterpret_cast<float4*>(Wt + k * HP + 4 * h4) = load_w4(p.W + (size_t)k * p.ldw + 4 * h4);
        }
        __syncthreads();
        for (int e = tid; e < E; e += NT) {
            const int n = e & 31, kg = (e >> 5) & 1, c = (e >> 6) % HT, m = e / (64 * HT);
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = Wt[(8 * m + 32 * kg + j) * HP + 32 * c + n];
            u32x4 p1, p2, p3;
            lin_split8(f, p1, p2, p3);
            Wp[e] = p1; Wp[E + e] = p2; Wp[2 * E + e] = p3;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): nothing from the preamble is pending inside the tile loop
    __syncthreads();

    GNM_SSTAMP(1)
    double st1[HT], st2[HT];
#pragma unroll
    for (int c = 0; c < HT; ++c) { st1[c] = 0.0; st2[c] = 0.0; }
    int tk = 0;               // (tuning builds: tile counter of the timeline stamps)
    auto do_tile = [&](int t) {
        GNM_SSTAMP(2 + 6 * min(tk, 9))
        const int r0 = t * 32;
        f32x16 acc[HT];
#pragma unroll
        for (int c = 0; c < HT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            float4 v = __builtin_bit_cast(float4, raw[j]);
            if (pro) {
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (p.pro_relu) {
                    v.x = gnm_relu(v.x); v.y = gnm_relu(v.y); v.z = gnm_relu(v.z); v.w = gnm_relu(v.w);
                }
            }
            *reinterpret_cast<float4*>(Xs + (lrow0 + j * RSTEP) * XS + 4 * c4) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        GNM_SSTAMP(3 + 6 * min(tk, 9))
        load_tile(t + tstride);                   // past the last tile: an empty descriptor, no memory traffic
        // row i, k = 32 h + 0..31 -> four A fragments (m: k = 8 m + 32 h + 0..7) x three planes, each split right in
        // front of its MFMAs (GNM_L64_SPLIT_AHEAD: all four first, as until round 4 -- 36 more registers, and the
        // splits cannot run in the MFMAs' shadow)
#ifdef GNM_L64_SPLIT_AHEAD
        u32x4 A1[4], A2[4], A3[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 v0 = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 8 * m);
            const float4 v1 = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 8 * m + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            lin_split8(f, A1[m], A2[m], A3[m]);
        }
        GNM_SSTAMP(4 + 6 * min(tk, 9))
#endif
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#ifdef GNM_L64_SPLIT_AHEAD
            const lin_bf16x8 a1 = __builtin_bit_cast(lin_bf16x8, A1[m]), a2 = __builtin_bit_cast(lin_bf16x8, A2[m]),
                             a3 = __builtin_bit_cast(lin_bf16x8, A3[m]);
#else
            u32x4 A1m, A2m, A3m;
            {
                const float4 v0 = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 8 * m);
                const float4 v1 = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 8 * m + 4);
                const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                lin_split8(f, A1m, A2m, A3m);
            }
            const lin_bf16x8 a1 = __builtin_bit_cast(lin_bf16x8, A1m), a2 = __builtin_bit_cast(lin_bf16x8, A2m),
                             a3 = __builtin_bit_cast(lin_bf16x8, A3m);
#endif
#pragma unroll
            for (int c = 0; c < HT; ++c) {
                const int e = (m * HT + c) * 64 + lane;
                const lin_bf16x8 b1 = __builtin_bit_cast(lin_bf16x8, Wp[e]), b2 = __builtin_bit_cast(lin_bf16x8, Wp[E + e]),
                                 b3 = __builtin_bit_cast(lin_bf16x8, Wp[2 * E + e]);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[c], 0, 0, 0);      // small terms first
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[c], 0, 0, 0);
            }
        }
        GNM_SSTAMP(5 + 6 * min(tk, 9))
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // A fragments are in registers before the staging image is reused
        // (only the launch's last tile can have rows past N: every other tile sums without the per-element row guard --
        //  two selects per element, a fifth of the tile's vector instructions)
        const bool full = r0 + 32 <= p.N;                  // wave-uniform
        float* const xo = Xs + (4 * h) * XS + i;
#pragma unroll
        for (int c = 0; c < HT; ++c) {
            float s1 = 0.f, s2 = 0.f;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float z = acc[c][r] + bias_r[c];
                    xo[((r & 3) + 8 * (r >> 2)) * XS + 32 * c] = z;
                    s1 += z;
                    s2 += z * z;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float z = acc[c][r] + bias_r[c];
                    xo[((r & 3) + 8 * (r >> 2)) * XS + 32 * c] = z;
                    if (r0 + lrow < p.N) {
                        s1 += z;
                        s2 += z * z;
                    }
                }
            }
            st1[c] += (double)s1;
            st2[c] += (double)s2;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        GNM_SSTAMP(6 + 6 * min(tk, 9))
        const __amdgpu_buffer_rsrc_t rz = gnm_tile_rsrc(p.Z + (size_t)r0 * p.ldz, (GNM_L64_ABLATE & 1) ? 0 : min(p.N - r0, 32), p.ldz, HP);
        // (idx = lane + 64 st -> row = lane / O4 + WSTEP st, chunk = lane % O4: one address per lane, the rest immediates)
        const float* const xi = Xs + (lane / O4) * XS + 4 * (lane % O4);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xi + st * WSTEP * XS);
            __builtin_amdgcn_raw_buffer_store_b128(v, rz, out_voff + st * out_step, 0, 0);     // (row step in the vector offset: see above)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GNM_SSTAMP(7 + 6 * min(tk, 9))
        ++tk;
    };

    if (t < ntiles) {
        do_tile(t);
        for (t += tstride; t < ntiles; t += tstride) do_tile(t);
    }
    GNM_SSTAMP(62)

    if (p.stats_partial) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);   // [NW waves][2][HP]
#pragma unroll
        for (int c = 0; c < HT; ++c) {
            const double a1 = st1[c] + __shfl_xor(st1[c], 32, 64);
            const double a2 = st2[c] + __shfl_xor(st2[c], 32, 64);
            if (h == 0) {
                red[(wave * 2 + 0) * HP + 32 * c + i] = a1;
                red[(wave * 2 + 1) * HP + 32 * c + i] = a2;
            }
        }
        __syncthreads();
        // three rows of partials per workgroup (waves 0-3, 4-7, 8-11): gnm_linear_grid(N) rows per launch, as always
        for (int idx = tid; idx < (NW / 4) * 2 * HP; idx += NT) {
            const int grp = idx / (2 * HP), rest = idx - grp * 2 * HP;
            const int which = rest / HP, col = rest - which * HP;
            const int row = blockIdx.x * (NW / 4) + grp;
            if (row >= p.stat_rows) continue;
            double s = 0.0;
            for (int w = 4 * grp; w < 4 * grp + 4; ++w) s += red[(w * 2 + which) * HP + col];
            p.stats_partial[((size_t)row * 2 + which) * p.H + col] = s;
        }
    }
}

template <int HT>
static int launch_lin_split(const LinArgs& a0, int grid, hipStream_t s) {
    LinArgs a = a0;
    a.stat_rows = grid;
    const int grid3 = (grid + kSplitWaves / 4 - 1) / (kSplitWaves / 4);
    constexpr int HP = HT * 32;
    constexpr int XS = (64 > HP ? 64 : HP) + 4;
    const size_t lds = (size_t)3 * 4 * HT * 64 * 16 + (size_t)kSplitWaves * 32 * XS * 4;
    if (lds > (size_t)kLdsBudget) return GNM_ERR_UNSUPPORTED;
    GNM_ALLOW_FULL_LDS((&gnm_lin_split_kernel<HT>));
    hipLaunchKernelGGL((gnm_lin_split_kernel<HT>), dim3(grid3), dim3(kSplitWaves * 64), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------
// Split-precision Linear for K = 128, H = 128 (round 3: every hidden Linear of BASELINE configs[3], forward and dgrad).
// The fp32-instruction kernels run this shape at 121 us per [256,000 x 128 x 128] launch = 69 TFLOP/s, 44 % of the
// fp32 matrix peak, while moving 2.1 TB/s: matrix-pipe bound.  The K = 64 kernel above does not stretch -- three weight
// planes of [128 x 128] are 98 KB and its per-wave fp32 staging image would be 17 KB: room for 3 waves -- so this one
// keeps the weight planes in LDS and stages the input ONE 16-k SLICE of a tile at a time through a 2.5 KB per-wave
// image: four lanes read a row's 64 contiguous bytes of the slice (16 rows per instruction, whole 64-byte sectors), the
// image is read back in operand order (lane (i, h): row i, k = 16 step + 8 h + 0..7), split in registers (optionally
// after the BatchNorm + ReLU prologue, whose vectors sit in LDS) and multiplied as six bf16 terms against the planes.
// (Round 3 loaded the operands straight from global memory, every lane its own row: 64 requests of 16 useful bytes per
//  instruction, each 128-byte line touched by eight instructions -- the launch ran at 6.8 B/clk per CU where the K = 64
//  kernel, whose loads are row-contiguous, gets 9.5: round 4.)  The next slice -- across tile boundaries too, under the
// epilogue's stores -- is requested before the current one is multiplied.
// Twelve waves per CU (one workgroup; 168 registers), tiles of 32 rows dealt to the waves as in the K = 64 kernel, so
// the launch writes the same gnm_linear_grid(N) rows of statistics partials.  The accumulators are stored as they stand
// (lane = column: 128-byte row segments, 4 bytes per lane).
// ---------------------------------------------------------------------------------
static constexpr int kSplit128Waves = 12;
#ifndef GNM_L128_ABLATE          // tuning builds (tools/build_variant.py -DGNM_L128_ABLATE=n): 1 = no output stores,
#define GNM_L128_ABLATE 0        // 2 = slices come from an empty descriptor (no input traffic)
#endif

template <bool MASKED>
__global__ void __launch_bounds__(kSplit128Waves * 64) gnm_lin_split128_kernel(const LinArgs p) {
    constexpr int K = 128, HT = 4, HP = 128, NW = kSplit128Waves, NT = NW * 64;
    constexpr int E = 2 * 4 * HT * 64;              // 16-byte operand entries per weight plane: [kk][m][c][lane]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* Wp = reinterpret_cast<u32x4*>(smem);                              // [3][E]
    float* psv = reinterpret_cast<float*>(smem + (size_t)3 * E * 16);        // [3][K]: prologue scale / shift, bias
    // column statistics of this wave, fp64, in LDS ([2][HP] per wave: 16 registers a lane could not spare -- with them
    // in registers the tile loop spilled 21)
    double* wst = reinterpret_cast<double*>(smem + (size_t)3 * E * 16 + 3 * K * 4) + (size_t)(threadIdx.x >> 6) * 2 * HP;
    // the wave's slice image: [32 rows][16 k + 4 floats of padding] (80-byte rows: the 32-byte operand reads of 16
    // consecutive rows fall on 64 distinct banks)
    constexpr int XS = 20;
    float* xs = reinterpret_cast<float*>(smem + (size_t)3 * E * 16 + 3 * K * 4 + (size_t)NW * 2 * HP * 8) +
                (size_t)(threadIdx.x >> 6) * 32 * XS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31;
    const int h = lane >> 5;
    const bool pro = p.pro_scale != nullptr;
    // weight planes: entry (step, c, lane = 32 kg + n) = W[h = 32 c + n][k = 16 step + 8 kg + 0..7]
    for (int e = tid; e < E; e += NT) {
        const int n = e & 31, kg = (e >> 5) & 1, c = (e >> 6) & 3, step = e >> 8;
        const int k0 = 16 * step + 8 * kg, hh = 32 * c + n;
        float f[8];
        if (!p.w_kmajor && (p.ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(p.W) & 15) == 0) {     // (kernel-uniform)
            // torch's layout: the entry's eight k are 32 contiguous bytes of row hh
            const float4 w0 = *reinterpret_cast<const float4*>(p.W + (size_t)hh * p.ldw + k0);
            const float4 w1 = *reinterpret_cast<const float4*>(p.W + (size_t)hh * p.ldw + k0 + 4);
            f[0] = w0.x; f[1] = w0.y; f[2] = w0.z; f[3] = w0.w; f[4] = w1.x; f[5] = w1.y; f[6] = w1.z; f[7] = w1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                f[j] = p.w_kmajor ? p.W[(size_t)(k0 + j) * p.ldw + hh] : p.W[(size_t)hh * p.ldw + k0 + j];
        }
        u32x4 p1, p2, p3;
        lin_split8(f, p1, p2, p3);
        Wp[e] = p1; Wp[E + e] = p2; Wp[2 * E + e] = p3;
    }
    for (int e = tid; e < K; e += NT) {
        psv[e] = pro ? p.pro_scale[e] : 1.f;
        psv[K + e] = pro ? p.pro_shift[e] : 0.f;
        psv[2 * K + e] = p.bias ? p.bias[e] : 0.f;           // (H == K == 128)
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): nothing from the preamble is pending inside the tile loop
    __syncthreads();

    const int gw = lin_first_tile(wave, NW / 4, p.stat_rows);
    const int ntiles = (p.N + 31) / 32;
    const int tstride = 4 * p.stat_rows;
    // a slice request: lane -> row (lane >> 2) (+ 16 for the second instruction), 16-byte chunk (lane & 3) of the 64 bytes
    const int ld_voff = ((lane >> 2) * p.ldx + 4 * (lane & 3)) * 4, ld_row16 = 16 * p.ldx * 4;
    float* const xs_w = xs + (lane >> 2) * XS + 4 * (lane & 3);
    const float* const xs_r = xs + i * XS + 8 * h;
    for (int e = lane; e < 2 * HP; e += 64) wst[e] = 0.0;       // (wave-private: no barrier needed)

    auto tile_rsrc_of = [&](int t) {          // (a tile past the end: an empty descriptor, every load returns zeros)
        const int r0 = min(t, ntiles - 1) * 32;
        return gnm_tile_rsrc(p.X + (size_t)r0 * p.ldx, (t < ntiles && !(GNM_L128_ABLATE & 2)) ? min(p.N - r0, 32) : 0, p.ldx, K);
    };
    // Slices are requested D steps ahead into a register ring (slot = slice index mod D): with one slice of 2 KB in
    // flight per wave a CU had 24 KB on its way -- 12 GB/s per CU at the ~2 us a loaded HBM round trip takes, the 3.7 TB/s
    // this kernel ran at; D = 4 (the plain form; the masked form has registers for 2) keeps up to 96 KB in flight.
    constexpr int D = MASKED ? 2 : 4;
    int t = gw;
    u32x4 ring0[D], ring1[D];
#pragma unroll
    for (int q = 0; q < D; ++q) { ring0[q] = u32x4{0u, 0u, 0u, 0u}; ring1[q] = ring0[q]; }
    if (t < ntiles) {
        const __amdgpu_buffer_rsrc_t rs0 = tile_rsrc_of(t);
#pragma unroll
        for (int q = 0; q < D; ++q) {
            ring0[q] = __builtin_amdgcn_raw_buffer_load_b128(rs0, ld_voff, q * 64, 0);
            ring1[q] = __builtin_amdgcn_raw_buffer_load_b128(rs0, ld_voff + ld_row16, q * 64, 0);
        }
    }
    // (the first tile is peeled below: the loop is then entered with the queue its back edge carries -- two slice loads
    //  followed by the epilogue's stores -- and the wait for the slice is a counted one instead of a drain of the stores)
    auto do_tile = [&](const int t) {
        const int r0 = t * 32;
        const int rows = min(p.N - r0, 32);
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc_of(t);
        const __amdgpu_buffer_rsrc_t rs_next = tile_rsrc_of(t + tstride);
        f32x16 acc[HT];
#pragma unroll
        for (int c = 0; c < HT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        // slice 0 (requested under the previous tile's epilogue) into the image; LDS operations of one wave execute in
        // order, so the image needs no second buffer and no barrier
        *reinterpret_cast<u32x4*>(xs_w) = ring0[0];
        *reinterpret_cast<u32x4*>(xs_w + 16 * XS) = ring1[0];
        // eight k steps of 16, unrolled (the ring slots are compile-time) with a scheduling fence per step: left alone,
        // the scheduler hoists every load and split of the 192-MFMA body (295 spilled registers in round 3)
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            __builtin_amdgcn_sched_barrier(0);
            {   // slice step + D -- of the next tile past the end of this one -- into the slot slice `step` has left
                const int sl = step + D;
                const __amdgpu_buffer_rsrc_t rq = sl < 8 ? rs : rs_next;
                ring0[step % D] = __builtin_amdgcn_raw_buffer_load_b128(rq, ld_voff, (sl & 7) * 64, 0);
                ring1[step % D] = __builtin_amdgcn_raw_buffer_load_b128(rq, ld_voff + ld_row16, (sl & 7) * 64, 0);
            }
            const float4 v0 = *reinterpret_cast<const float4*>(xs_r), v1 = *reinterpret_cast<const float4*>(xs_r + 4);
            float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (pro) {
                // (a clipped row read zeros and the affine map moves them: its products land in accumulator rows that
                //  are neither stored nor counted)
                const int k0 = 16 * step + 8 * h;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 sc = *reinterpret_cast<const float4*>(psv + k0 + 4 * q);
                    const float4 sh = *reinterpret_cast<const float4*>(psv + K + k0 + 4 * q);
                    f[4 * q + 0] = f[4 * q + 0] * sc.x + sh.x; f[4 * q + 1] = f[4 * q + 1] * sc.y + sh.y;
                    f[4 * q + 2] = f[4 * q + 2] * sc.z + sh.z; f[4 * q + 3] = f[4 * q + 3] * sc.w + sh.w;
                }
                if (p.pro_relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = gnm_relu(f[j]);
                }
            }
            u32x4 A1, A2, A3;
            lin_split8(f, A1, A2, A3);
            const lin_bf16x8 a1 = __builtin_bit_cast(lin_bf16x8, A1), a2 = __builtin_bit_cast(lin_bf16x8, A2),
                             a3 = __builtin_bit_cast(lin_bf16x8, A3);
#pragma unroll
            for (int c = 0; c < HT; ++c) {
                const int e = ((step * HT + c) << 6) + lane;
                const lin_bf16x8 b1 = __builtin_bit_cast(lin_bf16x8, Wp[e]), b2 = __builtin_bit_cast(lin_bf16x8, Wp[E + e]),
                                 b3 = __builtin_bit_cast(lin_bf16x8, Wp[2 * E + e]);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[c], 0, 0, 0);      // small terms first
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[c], 0, 0, 0);
            }
            if (step < 7) {       // the next slice into the image (behind this step's reads, in order)
                *reinterpret_cast<u32x4*>(xs_w) = ring0[(step + 1) % D];
                *reinterpret_cast<u32x4*>(xs_w + 16 * XS) = ring1[(step + 1) % D];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue: bias, column statistics, stores (lane = column; rows past N clipped by the descriptor) ----
        const __amdgpu_buffer_rsrc_t rz = gnm_tile_rsrc(p.Z + (size_t)r0 * p.ldz, (GNM_L128_ABLATE & 1) ? 0 : rows, p.ldz, HP);
        if constexpr (MASKED) {
            // masked dX: the values under the accumulators of the lower BatchNorm's input (lane = column: 128-byte row
            // pieces), one column block requested ahead of the one being finished
            const __amdgpu_buffer_rsrc_t rm = gnm_tile_rsrc(p.mZ + (size_t)r0 * p.ldmz, rows, p.ldmz, HP);
            // (one vector offset per lane and the (row, column block) part in the scalar offset: with a vector offset per
            //  element the compiler kept 128 of them across the tile loop -- 77 spilled registers)
            const unsigned mvo = (unsigned)((4 * h * p.ldmz + i) * 4), zvo = (unsigned)((4 * h * p.ldz + i) * 4);
            float zm[2][16];
            auto req = [&](float (&d)[16], int c) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    d[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rm, mvo, (((r & 3) + 8 * (r >> 2)) * p.ldmz + 32 * c) * 4, 0));
            };
            __builtin_amdgcn_sched_barrier(0);            // (hoisted into the product loop these requests spilled 77 registers)
            req(zm[0], 0);
#pragma unroll
            for (int c = 0; c < HT; ++c) {
                if (c + 1 < HT) req(zm[(c + 1) & 1], c + 1);
                __builtin_amdgcn_sched_barrier(0);
                const int col = 32 * c + i;
                const float msc = p.m_scale[col], msh = p.m_shift[col], mmu = p.m_mean[col];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float zv = zm[c & 1][r];
                    float g = acc[c][r];
                    if (!(zv * msc + msh > 0.f)) g = 0.f;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(g), rz, zvo, (((r & 3) + 8 * (r >> 2)) * p.ldz + 32 * c) * 4, 0);
                    if (lrow < rows) {
                        s1 += g;
                        s2 += g * (zv - mmu);
                    }
                }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (h == 0) {
                    wst[col] += (double)s1;
                    wst[HP + col] += (double)s2 * (double)p.m_rstd[col];      // sum G (Z - mean) rstd = sum G xhat
                }
            }
        } else {
        // (one vector offset per lane, the (row, column block) part in the scalar offset: a vector offset per store is 64
        //  registers the compiler keeps -- and spills -- across the tile loop)
        const unsigned zvo = (unsigned)((4 * h * p.ldz + i) * 4);
#pragma unroll
        for (int c = 0; c < HT; ++c) {
            float s1 = 0.f, s2 = 0.f;
            const float bz = psv[2 * K + 32 * c + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float z = acc[c][r] + bz;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(z), rz, zvo, (((r & 3) + 8 * (r >> 2)) * p.ldz + 32 * c) * 4, 0);
                if (lrow < rows) {
                    s1 += z;
                    s2 += z * z;
                }
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                wst[32 * c + i] += (double)s1;
                wst[HP + 32 * c + i] += (double)s2;
            }
        }
        }
    };
    if (t < ntiles) {
        do_tile(t);
        for (t += tstride; t < ntiles; t += tstride) do_tile(t);
    }

    if (p.stats_partial) {
        const double* red = reinterpret_cast<const double*>(smem + (size_t)3 * E * 16 + 3 * K * 4);   // [NW waves][2][HP]
        __syncthreads();
        for (int idx = tid; idx < 3 * 2 * HP; idx += NT) {        // three rows of partials per workgroup, as above
            const int grp = idx / (2 * HP), rest = idx - grp * 2 * HP;
            const int which = rest / HP, col = rest - which * HP;
            const int row = blockIdx.x * 3 + grp;
            if (row >= p.stat_rows) continue;
            double sacc = 0.0;
            for (int w = 4 * grp; w < 4 * grp + 4; ++w) sacc += red[(w * 2 + which) * HP + col];
            p.stats_partial[((size_t)row * 2 + which) * p.H + col] = sacc;
        }
    }
}

static int launch_lin_split128(const LinArgs& a0, int grid, hipStream_t s) {
    LinArgs a = a0;
    a.stat_rows = grid;
    const int grid3 = (grid + 2) / 3;
    const size_t lds = (size_t)3 * (2 * 4 * 4 * 64) * 16 + (size_t)3 * 128 * 4 + (size_t)kSplit128Waves * 2 * 128 * 8 +
                       (size_t)kSplit128Waves * 32 * 20 * 4;          // planes, prologue vectors, statistics, slice images
    if (a.mZ) {
        GNM_ALLOW_FULL_LDS((&gnm_lin_split128_kernel<true>));
        hipLaunchKernelGGL(gnm_lin_split128_kernel<true>, dim3(grid3), dim3(kSplit128Waves * 64), lds, s, a);
    } else {
        GNM_ALLOW_FULL_LDS((&gnm_lin_split128_kernel<false>));
        hipLaunchKernelGGL(gnm_lin_split128_kernel<false>, dim3(grid3), dim3(kSplit128Waves * 64), lds, s, a);
    }
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

static bool lin_no_split() {
#ifdef GNM_LIN_FORCE_NO_SPLIT          // variant builds for paired timing (tools/build_variant.py)
    return true;
#else
    static const bool v = gnm_env_int("GNM_LIN_NO_SPLIT", 0) != 0;     // A/B knob: keep the fp32 matrix instruction
    return v;
#endif
}
// the fused backward kernels' products alone (GNM_LIN_NO_SPLIT covers them too)
static bool linbwd_no_split() {
    static const bool v = gnm_env_int("GNM_LINBWD_NO_SPLIT", 0) != 0;
    return v || lin_no_split();
}

static size_t lin_lds_bytes(int K, int KC, int HT) {
    const int KP = ((K + KC - 1) / KC) * KC;
    size_t b = (size_t)KP * HT * 32 * 4 + (size_t)4 * 32 * (KC + 4) * 4;
    const size_t red = (size_t)4 * 2 * HT * 32 * 8;
    return b > red ? b : red;
}

template <int KC, int HT>
static int launch_lin(const LinArgs& a0, int grid, hipStream_t s) {
    LinArgs a = a0;
    size_t lds = lin_lds_bytes(a.K, KC, HT);
    if (lds > (size_t)kLdsBudget) return GNM_ERR_UNSUPPORTED;
    // room for the output staging image (a wide first layer, K = 400 one-hot, has none: 4-byte column stores then)
    const size_t wbytes = (size_t)(((a.K + KC - 1) / KC) * KC) * HT * 32 * 4 + (size_t)4 * 32 * (KC + 4) * 4;
    const size_t obytes = (size_t)4 * 32 * (HT * 32 + 4) * 4;
    a.stage_out = wbytes + obytes <= (size_t)kLdsBudget / 2 ? 1 : 0;      // only while two workgroups still share a CU
    if (a.stage_out && wbytes + obytes > lds) lds = wbytes + obytes;
    GNM_ALLOW_FULL_LDS((&gnm_lin_kernel<KC, HT>));
    hipLaunchKernelGGL((gnm_lin_kernel<KC, HT>), dim3(grid), dim3(256), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// Largest input width K that gnm_linear_fwd accepts for output width H (the weight [K x H] stays in LDS for the
// kernel's lifetime): what a caller checks before building a model around it.  0 when H itself is unsupported.
extern "C" int gnm_linear_max_k(int H) {
    if (H <= 0 || H > 128) return 0;
    const int HT = (H + 31) / 32;
    int k = 0;
    while (lin_lds_bytes(k + 64, 64, HT) <= (size_t)kLdsBudget) k += 64;
    return k;
}

// Blocks gnm_linear_fwd launches for N rows (= rows of stats_partial it writes).
extern "C" int gnm_linear_grid(int N) {
    const int ntiles = (N + 31) / 32;
    int g = (ntiles + 3) / 4;
    static const int cap = gnm_env_int("GNM_LIN_GRID", 768);   // tuning knob
    if (g > cap) g = cap;
    return g < 1 ? 1 : g;
}

// Z[N,H] = f(X)[N,K] * W^T + bias, f = optional fused affine(+ReLU) prologue; optional
// per-column (sum, sum of squares) partials for the following BatchNorm.
extern "C" int gnm_linear_fwd(const float* X, int ldx, const float* W, int ldw, int w_kmajor, const float* bias,
                              float* Z, int ldz, int N, int K, int H, const float* pro_scale,
                              const float* pro_shift, int pro_relu, double* stats_partial, void* stream) {
    if (N <= 0) return GNM_OK;
    if (K <= 0 || H <= 0 || H > 128) return GNM_ERR_BAD_ARG;
    LinArgs a;
    a.X = X; a.W = W; a.bias = bias; a.Z = Z; a.pro_scale = pro_scale; a.pro_shift = pro_shift;
    a.stats_partial = stats_partial; a.ldx = ldx; a.ldw = ldw; a.ldz = ldz; a.N = N; a.K = K; a.H = H;
    a.w_kmajor = w_kmajor; a.pro_relu = pro_relu;
    a.stat_rows = 0; a.stage_out = 0;
    a.mZ = nullptr; a.m_scale = a.m_shift = a.m_mean = a.m_rstd = nullptr; a.ldmz = 0;
#ifdef GNM_LIN_TUNING
    a.stamps = g_lin_stamps;
#else
    a.stamps = nullptr;
#endif
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = gnm_linear_grid(N);
    const int HT = (H + 31) / 32;
    const int kc = K <= 8 ? 8 : (K <= 16 ? 16 : (K <= 32 ? 32 : 64));
    // pipelined variant: aligned rows in and out, no partial chunks or partial output tiles
    const bool aligned = ((ldx & 3) == 0) && ((ldz & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(Z) & 15) == 0) &&
                         (!pro_scale || (((reinterpret_cast<uintptr_t>(pro_scale) | reinterpret_cast<uintptr_t>(pro_shift)) & 15) == 0));
    if (aligned && (H % 32) == 0 && (K % 32) == 0 && !lin_force_generic()) {
        const int kcf = (K % 64) == 0 ? 64 : 32;
        int rc = GNM_ERR_UNSUPPORTED;
        // one chunk per row and descriptors that fit 32-bit offsets: the streaming kernel
        const bool small_ld = (long long)ldx * 32 * 4 < (1LL << 31) && (long long)ldz * 32 * 4 < (1LL << 31);
        // K = H = 64: the split-precision kernel
        if (K == 64 && HT == 2 && small_ld && !lin_no_stream() && !lin_no_split()) {
            rc = launch_lin_split<2>(a, grid, s);
            if (rc != GNM_ERR_UNSUPPORTED) return rc;
        }
        // K = H = 128 (configs[3]): its own split-precision kernel
        if (K == 128 && H == 128 && small_ld && !lin_no_split()) return launch_lin_split128(a, grid, s);
        if ((K == 32 || K == 64) && small_ld && !lin_no_stream()) {
#define GNM_LINS_CASE(KC_, HT_) if (K == KC_ && HT == HT_) rc = launch_lin_stream<KC_, HT_>(a, grid, s);
            GNM_LINS_CASE(32, 1) GNM_LINS_CASE(32, 2) GNM_LINS_CASE(32, 3) GNM_LINS_CASE(32, 4)
            GNM_LINS_CASE(64, 1) GNM_LINS_CASE(64, 2) GNM_LINS_CASE(64, 3) GNM_LINS_CASE(64, 4)
#undef GNM_LINS_CASE
            if (rc != GNM_ERR_UNSUPPORTED) return rc;
        }
#define GNM_LINF_CASE(KC_, HT_) if (kcf == KC_ && HT == HT_) rc = launch_lin_fast<KC_, HT_>(a, grid, s);
        GNM_LINF_CASE(32, 1) GNM_LINF_CASE(32, 2) GNM_LINF_CASE(32, 3) GNM_LINF_CASE(32, 4)
        GNM_LINF_CASE(64, 1) GNM_LINF_CASE(64, 2) GNM_LINF_CASE(64, 3) GNM_LINF_CASE(64, 4)
#undef GNM_LINF_CASE
        if (rc != GNM_ERR_UNSUPPORTED) return rc;      // too large for LDS: fall through to the generic kernel
    }
#define GNM_LIN_CASE(KC_, HT_) if (kc == KC_ && HT == HT_) return launch_lin<KC_, HT_>(a, grid, s);
    GNM_LIN_CASE(8, 1) GNM_LIN_CASE(8, 2) GNM_LIN_CASE(8, 3) GNM_LIN_CASE(8, 4)
    GNM_LIN_CASE(16, 1) GNM_LIN_CASE(16, 2) GNM_LIN_CASE(16, 3) GNM_LIN_CASE(16, 4)
    GNM_LIN_CASE(32, 1) GNM_LIN_CASE(32, 2) GNM_LIN_CASE(32, 3) GNM_LIN_CASE(32, 4)
    GNM_LIN_CASE(64, 1) GNM_LIN_CASE(64, 2) GNM_LIN_CASE(64, 3) GNM_LIN_CASE(64, 4)
#undef GNM_LIN_CASE
    return GNM_ERR_UNSUPPORTED;
}

// dX = dZ W for a K = H = 128 Linear whose input came through BatchNorm + ReLU (mlp.py:48), with that ReLU's mask and
// that BatchNorm's backward sums taken on the way out (replaces gnm_linear_fwd(w_kmajor = 1) + gnm_bn_relu_bwd_stats for
// the inner BatchNorms of an H = 128 model; the K = H = 64 shapes have gnm_linear_bwd_fused):
//     G[n, k] = (dZ W)[n, k] if relu(mZ[n, k] m_scale[k] + m_shift[k]) > 0 else 0
//     stats_partial[row][0][k] = partial sum G,  [1][k] = partial sum G (mZ - m_mean) m_rstd     (gnm_linear_grid(N) rows)
// dZ [N, H = 128], W [H][K = 128] row-major (the Linear's weight), G [N, K].  GNM_ERR_UNSUPPORTED for other shapes.
extern "C" int gnm_linear_dgrad_masked(const float* dZ, int ldd, const float* W, int ldw, float* G, int ldg, int N, int K,
                                       int H, const float* mZ, int ldmz, const float* m_scale, const float* m_shift,
                                       const float* m_mean, const float* m_rstd, double* stats_partial, void* stream) {
    if (N <= 0) return GNM_OK;
    if (K != 128 || H != 128 || lin_no_split() || lin_force_generic()) return GNM_ERR_UNSUPPORTED;
    if (!dZ || !W || !G || !mZ || !m_scale || !m_shift || !m_mean || !m_rstd || !stats_partial) return GNM_ERR_BAD_ARG;
    if ((ldd & 3) || (reinterpret_cast<uintptr_t>(dZ) & 15)) return GNM_ERR_UNSUPPORTED;
    if ((long long)(ldd > ldg ? (ldd > ldmz ? ldd : ldmz) : (ldg > ldmz ? ldg : ldmz)) * 32 * 4 >= (1LL << 31)) return GNM_ERR_UNSUPPORTED;
    LinArgs a;
    a.X = dZ; a.W = W; a.bias = nullptr; a.Z = G; a.pro_scale = nullptr; a.pro_shift = nullptr;
    a.stats_partial = stats_partial; a.ldx = ldd; a.ldw = ldw; a.ldz = ldg; a.N = N; a.K = H; a.H = K;
    a.w_kmajor = 1; a.pro_relu = 0; a.stamps = nullptr; a.stat_rows = 0; a.stage_out = 0;
    a.mZ = mZ; a.m_scale = m_scale; a.m_shift = m_shift; a.m_mean = m_mean; a.m_rstd = m_rstd; a.ldmz = ldmz;
    return launch_lin_split128(a, gnm_linear_grid(N), reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------
// weight gradient: dW[h][k] = sum_n dZ[n][h] * f(X)[n][k],  db[h] = sum_n dZ[n][h]
// MFMA 32x32x2 with the batch row as the contraction index: lane (i, half) feeds
// A = dZ[n+half][32*ti+i] and B = f(X)[n+half][32*tj+i] straight from global memory
// (128-B contiguous per half-wave).  Each block reduces its row range to one
// [H x KW] partial; gnm_reduce_partials sums the partials in a fixed order.
// ------------------------------------------------------------------------------
struct WgArgs {
    const float* dZ;
    const float* X;
    const float* pro_scale;  // [K] or null
    const float* pro_shift;
    float* partial;          // [gridDim.x][H*kw + H]
    int ldd, ldx;
    int N, H, K;
    int k0, kw;              // column window of X handled by this launch (kw <= 128)
    int rows_per_block;      // even
    int pro_relu;
};

template <int WI, int WJ, int QI, int QJ>
__global__ void __launch_bounds__(256) gnm_wgrad_kernel(const WgArgs p) {
    constexpr int NQ = QI * QJ;          // quadrants of the output, one per wave group
    constexpr int RSPLIT = 4 / NQ;       // waves sharing a quadrant split the rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dump = reinterpret_cast<float*>(smem);   // [4][WI*WJ][16][64] (+ db [4][WI][64])
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int q = wave % NQ, rs = wave / NQ;
    const int qi = q / QJ, qj = q % QJ;

    f32x16 acc[WI][WJ];
#pragma unroll
    for (int a = 0; a < WI; ++a)
#pragma unroll
        for (int b = 0; b < WJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float dbacc[WI];
#pragma unroll
    for (int a = 0; a < WI; ++a) dbacc[a] = 0.f;

    int hcol[WI], kcol[WJ];
    bool hok[WI], kok[WJ];
    float sc[WJ], sh[WJ];
#pragma unroll
    for (int a = 0; a < WI; ++a) { hcol[a] = 32 * (qi * WI + a) + i; hok[a] = hcol[a] < p.H; }
#pragma unroll
    for (int b = 0; b < WJ; ++b) {
        const int kl = 32 * (qj * WJ + b) + i;       // column inside the window
        kcol[b] = p.k0 + kl;
        kok[b] = (kl < p.kw) && (kcol[b] < p.K);
        sc[b] = (p.pro_scale && kok[b]) ? p.pro_scale[kcol[b]] : 1.f;
        sh[b] = (p.pro_scale && kok[b]) ? p.pro_shift[kcol[b]] : 0.f;
    }

    const int rb = blockIdx.x * p.rows_per_block;
    const int re = min(p.N, rb + p.rows_per_block);
#pragma unroll 4
    for (int n0 = rb + 2 * rs; n0 < re; n0 += 2 * RSPLIT) {
        const int n = n0 + h;
        const bool rok = n < re;
        float av[WI], bv[WJ];
#pragma unroll
        for (int a = 0; a < WI; ++a) av[a] = (rok && hok[a]) ? p.dZ[(size_t)n * p.ldd + hcol[a]] : 0.f;
#pragma unroll
        for (int b = 0; b < WJ; ++b) {
            float x = 0.f;
            if (rok && kok[b]) {
                x = p.X[(size_t)n * p.ldx + kcol[b]];
                if (p.pro_scale) {
                    x = x * sc[b] + sh[b];
                    if (p.pro_relu) x = gnm_relu(x);
                }
            }
            bv[b] = x;
        }
#pragma unroll
        for (int a = 0; a < WI; ++a) {
            dbacc[a] += av[a];
#pragma unroll
            for (int b = 0; b < WJ; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }

    // ---- dump per-wave accumulators, then combine waves in a fixed order ---------
    constexpr int TILE = 16 * 64;
    float* mine = dump + (size_t)wave * WI * WJ * TILE;
#pragma unroll
    for (int a = 0; a < WI; ++a)
#pragma unroll
        for (int b = 0; b < WJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(a * WJ + b) * TILE + r * 64 + lane] = acc[a][b][r];
    float* dbdump = dump + (size_t)4 * WI * WJ * TILE;   // [4][WI][64]
#pragma unroll
    for (int a = 0; a < WI; ++a) dbdump[(wave * WI + a) * 64 + lane] = dbacc[a];
    __syncthreads();

    float* out = p.partial + (size_t)blockIdx.x * ((size_t)p.H * p.kw + p.H);
    for (int idx = tid; idx < NQ * WI * WJ * TILE; idx += 256) {
        const int qq = idx / (WI * WJ * TILE);
        const int rem = idx - qq * (WI * WJ * TILE);
        const int ab = rem / TILE;
        const int rl = rem - ab * TILE;
        const int r = rl >> 6, ln = rl & 63;
        const int a = ab / WJ, b = ab - a * WJ;
        const int row = 32 * ((qq / QJ) * WI + a) + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int colw = 32 * ((qq % QJ) * WJ + b) + (ln & 31);
        if (row < p.H && colw < p.kw) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < RSPLIT; ++w) s += dump[(size_t)(w * NQ + qq) * WI * WJ * TILE + rem];
            out[(size_t)row * p.kw + colw] = s;
        }
    }
    // bias gradient: only the qj == 0 quadrants hold each dZ column exactly once
    for (int idx = tid; idx < QI * WI * 32; idx += 256) {
        const int qqi = idx / (WI * 32);
        const int rem = idx - qqi * (WI * 32);
        const int a = rem >> 5, ii = rem & 31;
        const int hh = 32 * (qqi * WI + a) + ii;
        if (hh < p.H) {
            float s = 0.f;
            const int qq = qqi * QJ;     // qj == 0
#pragma unroll
            for (int w = 0; w < RSPLIT; ++w) {
                const int wv = w * NQ + qq;
                s += dbdump[(wv * WI + a) * 64 + ii] + dbdump[(wv * WI + a) * 64 + 32 + ii];
            }
            out[(size_t)p.H * p.kw + hh] = s;
        }
    }
}

// Pipelined wgrad for H % 32 == 0 and K-window % 32 == 0: no column guards, rows clamped on
// load and zeroed by select, and U row-pairs (U*(WI+WJ) loads) in flight before their MFMAs.
// The generic kernel above waits for every pair's 4 loads before its 4 MFMAs.
template <int WI, int WJ, int QI, int QJ>
__global__ void __launch_bounds__(256) gnm_wgrad_fast_kernel(const WgArgs p) {
    constexpr int NQ = QI * QJ;
    constexpr int RSPLIT = 4 / NQ;
    constexpr int U = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dump = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int q = wave % NQ, rs = wave / NQ;
    const int qi = q / QJ, qj = q % QJ;

    f32x16 acc[WI][WJ];
#pragma unroll
    for (int a = 0; a < WI; ++a)
#pragma unroll
        for (int b = 0; b < WJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float dbacc[WI];
    int hcol[WI], kcol[WJ];
    bool kok[WJ];                 // columns past the window / K are clamped on load and zeroed by select
    float sc[WJ], sh[WJ];
#pragma unroll
    for (int a = 0; a < WI; ++a) { dbacc[a] = 0.f; hcol[a] = 32 * (qi * WI + a) + i; }
#pragma unroll
    for (int b = 0; b < WJ; ++b) {
        const int kl = 32 * (qj * WJ + b) + i;
        kok[b] = (kl < p.kw) && (p.k0 + kl < p.K);
        kcol[b] = min(p.k0 + kl, p.K - 1);
        sc[b] = p.pro_scale ? p.pro_scale[kcol[b]] : 1.f;
        sh[b] = p.pro_scale ? p.pro_shift[kcol[b]] : 0.f;
    }
    const int rb = blockIdx.x * p.rows_per_block;
    const int re = min(p.N, rb + p.rows_per_block);
    const int nlast = p.N - 1;
    for (int n0 = rb + 2 * rs; n0 < re; n0 += 2 * RSPLIT * U) {
        float av[U][WI], bv[U][WJ];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + u * 2 * RSPLIT + h;
            const int nc = min(n, nlast);
#pragma unroll
            for (int a = 0; a < WI; ++a) av[u][a] = p.dZ[(size_t)nc * p.ldd + hcol[a]];
#pragma unroll
            for (int b = 0; b < WJ; ++b) bv[u][b] = p.X[(size_t)nc * p.ldx + kcol[b]];
        }
        __builtin_amdgcn_sched_barrier(0);     // keep all U*(WI+WJ) loads ahead of the MFMAs
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool rok = (n0 + u * 2 * RSPLIT + h) < re;
#pragma unroll
            for (int b = 0; b < WJ; ++b) {
                float x = bv[u][b];
                if (p.pro_scale) {
                    x = x * sc[b] + sh[b];
                    if (p.pro_relu) x = gnm_relu(x);
                }
                bv[u][b] = (rok && kok[b]) ? x : 0.f;
            }
#pragma unroll
            for (int a = 0; a < WI; ++a) {
                const float d = rok ? av[u][a] : 0.f;
                dbacc[a] += d;
#pragma unroll
                for (int b = 0; b < WJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(d, bv[u][b], acc[a][b], 0, 0, 0);
            }
        }
    }

    constexpr int TILE = 16 * 64;
    float* mine = dump + (size_t)wave * WI * WJ * TILE;
#pragma unroll
    for (int a = 0; a < WI; ++a)
#pragma unroll
        for (int b = 0; b < WJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(a * WJ + b) * TILE + r * 64 + lane] = acc[a][b][r];
    float* dbdump = dump + (size_t)4 * WI * WJ * TILE;
#pragma unroll
    for (int a = 0; a < WI; ++a) dbdump[(wave * WI + a) * 64 + lane] = dbacc[a];
    __syncthreads();

    float* out = p.partial + (size_t)blockIdx.x * ((size_t)p.H * p.kw + p.H);
    for (int idx = tid; idx < NQ * WI * WJ * TILE; idx += 256) {
        const int qq = idx / (WI * WJ * TILE);
        const int rem = idx - qq * (WI * WJ * TILE);
        const int ab = rem / TILE;
        const int rl = rem - ab * TILE;
        const int r = rl >> 6, ln = rl & 63;
        const int a = ab / WJ, b = ab - a * WJ;
        const int row = 32 * ((qq / QJ) * WI + a) + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int colw = 32 * ((qq % QJ) * WJ + b) + (ln & 31);
        if (row < p.H && colw < p.kw) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < RSPLIT; ++w) s += dump[(size_t)(w * NQ + qq) * WI * WJ * TILE + rem];
            out[(size_t)row * p.kw + colw] = s;
        }
    }
    for (int idx = tid; idx < QI * WI * 32; idx += 256) {
        const int qqi = idx / (WI * 32);
        const int rem = idx - qqi * (WI * 32);
        const int a = rem >> 5, ii = rem & 31;
        const int hh = 32 * (qqi * WI + a) + ii;
        if (hh < p.H) {
            float s = 0.f;
            const int qq = qqi * QJ;
#pragma unroll
            for (int w = 0; w < RSPLIT; ++w) {
                const int wv = w * NQ + qq;
                s += dbdump[(wv * WI + a) * 64 + ii] + dbdump[(wv * WI + a) * 64 + 32 + ii];
            }
            out[(size_t)p.H * p.kw + hh] = s;
        }
    }
}

template <int WI, int WJ, int QI, int QJ>
static int launch_wgrad_fast(const WgArgs& a, int grid, hipStream_t s) {
    const size_t lds = ((size_t)4 * WI * WJ * 1024 + (size_t)4 * WI * 64) * 4;
    GNM_ALLOW_FULL_LDS((&gnm_wgrad_fast_kernel<WI, WJ, QI, QJ>));
    hipLaunchKernelGGL((gnm_wgrad_fast_kernel<WI, WJ, QI, QJ>), dim3(grid), dim3(256), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

template <int WI, int WJ, int QI, int QJ>
static int launch_wgrad(const WgArgs& a, int grid, hipStream_t s) {
    const size_t lds = ((size_t)4 * WI * WJ * 1024 + (size_t)4 * WI * 64) * 4;
    GNM_ALLOW_FULL_LDS((&gnm_wgrad_kernel<WI, WJ, QI, QJ>));
    hipLaunchKernelGGL((gnm_wgrad_kernel<WI, WJ, QI, QJ>), dim3(grid), dim3(256), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------
// K = H = 128 weight gradient on the bf16 matrix pipe (round 3: the hidden Linears of BASELINE configs[3]; the fp32
// instruction kernel above runs this shape at 124 us per 256,000 rows with one wave per SIMD).  The contraction index
// is the batch row, so an operand fragment is "eight consecutive rows of one column" per lane: exactly what a 4-byte
// load with lane = column delivers (128 contiguous bytes per half-wave), no transpose through LDS.  Both operands are
// split in registers into three exact bf16 planes (see gnm_lin_split_kernel) and six terms per 16 rows are issued.
// Eight waves per workgroup: wave (q, rs) owns the 64 x 64 quadrant q of dW and every second 16-row group of the
// workgroup's row range; the two row halves are added in a fixed order and the workgroup writes the same
// [H * kw + H] partial the fp32 kernels write, so gnm_reduce_partials_kernel and the workspace are unchanged.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) gnm_wgrad_split128_kernel(const WgArgs p) {
    constexpr int TILE = 16 * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dump = reinterpret_cast<float*>(smem);                 // [8 waves][4 tiles][16][64] + [8][2][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const int q = wave & 3, rs = wave >> 2;
    const int qi = q >> 1, qj = q & 1;
    const long long row0 = (long long)blockIdx.x * p.rows_per_block;
    const long long rows = min((long long)p.N - row0, (long long)p.rows_per_block);
    const __amdgpu_buffer_rsrc_t rd = gnm_tile_rsrc(p.dZ + row0 * p.ldd, rows, p.ldd, 128);
    const __amdgpu_buffer_rsrc_t rx = gnm_tile_rsrc(p.X + row0 * p.ldx + p.k0, rows, p.ldx, 128);
    int dvo[8], xvo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        dvo[j] = ((8 * hh + j) * p.ldd + 64 * qi + i) * 4;
        xvo[j] = ((8 * hh + j) * p.ldx + 64 * qj + i) * 4;
    }
    float psc[2], psh[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        psc[b] = p.pro_scale ? p.pro_scale[p.k0 + 64 * qj + 32 * b + i] : 1.f;
        psh[b] = p.pro_scale ? p.pro_shift[p.k0 + 64 * qj + 32 * b + i] : 0.f;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float dbs[2] = {0.f, 0.f};
    const int ngroups = rows > 0 ? (int)((rows + 15) / 16) : 0;

    // a group past the end reads zeros (the descriptor's range check covers the scalar offset: see gnm_tile_rsrc)
    auto load = [&](float (&d)[2][8], float (&x)[2][8], int g) {
        const int sd = g * 16 * p.ldd * 4, sx = g * 16 * p.ldx * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int a = 0; a < 2; ++a) d[a][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dvo[j] + 128 * a, sd, 0));
#pragma unroll
            for (int b = 0; b < 2; ++b) x[b][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xvo[j] + 128 * b, sx, 0));
        }
    };
    auto compute = [&](float (&d)[2][8], float (&x)[2][8]) {
        if (p.pro_scale) {
            // (a clipped row read zero and the affine map moves it: it meets a zero of dZ)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = x[b][j] * psc[b] + psh[b];
                    x[b][j] = p.pro_relu ? gnm_relu(v) : v;
                }
        }
        lin_bf16x8 dp[2][3], xp[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            dbs[a] += ((d[a][0] + d[a][1]) + (d[a][2] + d[a][3])) + ((d[a][4] + d[a][5]) + (d[a][6] + d[a][7]));
            u32x4 p1, p2, p3;
            lin_split8(d[a], p1, p2, p3);
            dp[a][0] = __builtin_bit_cast(lin_bf16x8, p1); dp[a][1] = __builtin_bit_cast(lin_bf16x8, p2);
            dp[a][2] = __builtin_bit_cast(lin_bf16x8, p3);
            lin_split8(x[a], p1, p2, p3);
            xp[a][0] = __builtin_bit_cast(lin_bf16x8, p1); xp[a][1] = __builtin_bit_cast(lin_bf16x8, p2);
            xp[a][2] = __builtin_bit_cast(lin_bf16x8, p3);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][0], xp[b][2], acc[a][b], 0, 0, 0);   // small terms first
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][2], xp[b][0], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][1], xp[b][1], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][0], xp[b][1], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][1], xp[b][0], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][0], xp[b][0], acc[a][b], 0, 0, 0);
            }
    };
    // two register images: group g + 2 is requested before group g is split and multiplied
    float dA_[2][8], xA_[2][8], dB_[2][8], xB_[2][8];
    load(dA_, xA_, rs);
#pragma nounroll
    for (int g = rs; g < ngroups; g += 4) {
        load(dB_, xB_, g + 2);
        compute(dA_, xA_);
        load(dA_, xA_, g + 4);
        compute(dB_, xB_);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // (the last, unused request)

    float* mine = dump + (size_t)wave * 4 * TILE;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(a * 2 + b) * TILE + r * 64 + lane] = acc[a][b][r];
    float* dbdump = dump + (size_t)8 * 4 * TILE;
#pragma unroll
    for (int a = 0; a < 2; ++a) dbdump[(wave * 2 + a) * 64 + lane] = dbs[a];
    __syncthreads();
    float* out = p.partial + (size_t)blockIdx.x * ((size_t)p.H * p.kw + p.H);
    for (int idx = tid; idx < 4 * 4 * TILE; idx += 512) {
        const int qq = idx / (4 * TILE);
        const int rem = idx - qq * (4 * TILE);
        const int ab = rem / TILE;
        const int rl = rem - ab * TILE;
        const int r = rl >> 6, ln = rl & 63;
        const int row = 64 * (qq >> 1) + 32 * (ab >> 1) + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int col = 64 * (qq & 1) + 32 * (ab & 1) + (ln & 31);
        out[(size_t)row * p.kw + col] = dump[(size_t)qq * 4 * TILE + rem] + dump[(size_t)(4 + qq) * 4 * TILE + rem];
    }
    for (int idx = tid; idx < 128; idx += 512) {
        const int qqi = idx >> 6, a = (idx >> 5) & 1, ii = idx & 31;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int wv = 4 * w + 2 * qqi;       // the quadrant with qj = 0 of this row half
            s += dbdump[(wv * 2 + a) * 64 + ii] + dbdump[(wv * 2 + a) * 64 + 32 + ii];
        }
        out[(size_t)p.H * p.kw + idx] = s;
    }
}

static int launch_wgrad_split128(const WgArgs& a, int grid, hipStream_t s) {
    const size_t lds = ((size_t)8 * 4 * 1024 + (size_t)8 * 2 * 64) * 4;
    GNM_ALLOW_FULL_LDS((&gnm_wgrad_split128_kernel));
    hipLaunchKernelGGL(gnm_wgrad_split128_kernel, dim3(grid), dim3(512), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

extern "C" int gnm_wgrad_grid(int N) {
    int g = (N + 511) / 512;            // >= 512 rows per block
    if (g > 256) g = 256;               // one block per CU: fewer partials to reduce
    return g < 1 ? 1 : g;
}

// Floats of workspace gnm_linear_wgrad needs: grid * (H*min(K,128) + H).
extern "C" long long gnm_wgrad_workspace_floats(int N, int H, int K) {
    const int kw = K < 128 ? K : 128;
    return (long long)gnm_wgrad_grid(N) * ((long long)H * kw + H);
}

// Second stage: sum the per-block partials in a fixed order and scatter the
// [H x kw] window into dW (leading dimension ldw, column offset k0) and db.
// 32 output elements per workgroup, 32 thread groups each summing every 32nd partial (so the
// ~8 MB of partials are streamed by ~130 workgroups instead of 17), combined through LDS.
__global__ void __launch_bounds__(1024) gnm_reduce_partials_kernel(const float* __restrict__ partial, int nblk,
                                                                   long long stride, int H, int kw, int k0,
                                                                   float* __restrict__ dW, int ldw,
                                                                   float* __restrict__ db) {
    __shared__ float red[32][32];
    const int tid = threadIdx.x;
    const int grp = tid >> 5, ngrp = (int)blockDim.x >> 5;       // 32 groups of 32 lanes: group g sums partials g, g+32, ...
    const int e = blockIdx.x * 32 + (tid & 31);
    const int count = H * kw + H;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < count) {
        int b = grp;
        for (; b + 3 * ngrp < nblk; b += 4 * ngrp) {       // 4 independent loads in flight, fixed summation order
            const float v0 = partial[(size_t)(b + 0 * ngrp) * stride + e], v1 = partial[(size_t)(b + 1 * ngrp) * stride + e];
            const float v2 = partial[(size_t)(b + 2 * ngrp) * stride + e], v3 = partial[(size_t)(b + 3 * ngrp) * stride + e];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; b < nblk; b += ngrp) s0 += partial[(size_t)b * stride + e];
    }
    red[grp][tid & 31] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (tid < 32 && e < count) {
        float s = 0.f;
        for (int g = 0; g < ngrp; ++g) s += red[g][tid];
        if (e < H * kw) {
            const int row = e / kw, col = e - row * kw;
            dW[(size_t)row * ldw + k0 + col] = s;
        } else if (db && k0 == 0) {
            db[e - H * kw] = s;
        }
    }
}

// ------------------------------------------------------------------------------
// Fused backward of one Linear that is followed by a train/eval BatchNorm (mlp.py:48,
// graphcnn.py:163): replaces  gnm_bn_bwd_apply -> gnm_linear_wgrad -> gnm_linear_fwd(dgrad)
// (10 passes over [N,H] arrays) by ONE pass that reads G (masked incoming gradient), Z (this
// Linear's output = the BatchNorm input) and X (this Linear's input) and writes dX:
//     dZ = cA * (G - m1 - xhat * m2),  xhat = (Z - mean) * rstd        (BatchNorm backward)
//     dX = dZ W                                                          (64 MFMAs per 32 rows)
//     dW = dZ^T f(X),  db = sum dZ                                       (64 MFMAs per 32 rows)
// The dZ tile lives only in the wave's LDS staging image: dgrad reads it row-wise
// (ds_read_b128 A fragments), wgrad column-wise (ds_read_b32, lane = column).  f(X) operands
// come straight from global memory (128 B per half-wave), all loads of a tile are issued
// before its MFMAs.  K, H in {32, 64} (accumulators: 32*K/32 + 16*(H/32)*(K/32) registers), or K < 32 (NARROW).
// ------------------------------------------------------------------------------
struct LbArgs {
    const float* G; const float* Z; const float* X; const float* W;
    const float* mean; const float* rstd; const float* cA; const float* m1; const float* m2;
    const float* pro_scale; const float* pro_shift;
    float* dA; float* partial;
    // optional: dA is the gradient arriving at relu(bn_lo(sZ)) -- apply that ReLU mask to it on the way
    // out and reduce the lower BatchNorm's backward sums (replaces gnm_bn_relu_bwd_stats for it)
    const float* sZ; const float* s_scale; const float* s_shift; const float* s_mean; const float* s_rstd;
    double* s_partial;         // [gridDim.x][2][K]
    int ldg, ldz, ldx, ldw, lda, ldsz;
    int N, K, H;
    int pro_relu;
    unsigned long long* stamps;   // tuning builds only, as LinArgs::stamps
    const float* bias;            // gnm_linear_bwd_rz_kernel only: Z is not read but recomputed as f(X) W^T + bias
    int part_rows;                // ... and the rows of `partial` / `s_partial` its launch writes (two per workgroup)
};

// NARROW: K < 32 (the first Linear of layer 0, K = F0): one zero-padded 32-column tile, guarded scalar
// accesses to X / W / dX, which are small next to the [N,H] streams G and Z.
// SAMEZ (with STATS): the lower BatchNorm's input sZ IS this Linear's input X and its scale/shift ARE the prologue
// (the second Linear of every MLP) -- the ReLU mask and the BatchNorm sums are then taken from the X values the
// weight-gradient product already holds in registers: the wgrad contraction index is permuted so that a lane's X
// rows are its dX accumulator rows, and no second pass over that array is made.
template <int KT, int HT, bool STATS, bool NARROW = false, bool SAMEZ = false, bool SPLITD = false>
__global__ void __launch_bounds__(256, 2) gnm_linear_bwd_fused_kernel(const LbArgs p) {   // 2 waves/SIMD: <= 256 registers
    static_assert(!SPLITD || (HT == 2 && ((SAMEZ && KT == 2) || NARROW)), "SPLITD: H = 64, the SAMEZ K = 64 form or the narrow one");
    static_assert(!NARROW || (KT == 1 && !STATS), "narrow K: one tile, no lower BatchNorm");
    static_assert(!SAMEZ || STATS, "SAMEZ is a STATS variant");
    constexpr int KP = KT * 32, HP = HT * 32;
    constexpr int XS = (KP > HP ? KP : HP) + 4;
    constexpr int H4 = HP / 4;                    // float4 per dZ row
    constexpr int NLD = (32 * H4) / 64;           // float4 loads per lane for a [32][HP] tile
    constexpr int RSTEP = 64 / H4;
    constexpr int KH = HP / 2;                    // dgrad MFMA steps (contraction over H)
    constexpr int O4 = KP / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // SPLITD (the SAMEZ K = H = 64 form): the dgrad product runs on the bf16 matrix pipe with dZ and W split into three
    // exact bf16 planes each (see gnm_lin_split_kernel); the weight image is then the three operand planes (24 KB)
    constexpr int EW = 4 * KT * 64;
    float* Wt = reinterpret_cast<float*>(smem);                   // [HP][KP]: W itself (contraction index first)
    u32x4* Wp = reinterpret_cast<u32x4*>(smem);                   // SPLITD: [3][EW] operand entries instead
    float* Xs_all = SPLITD ? reinterpret_cast<float*>(smem + (size_t)3 * EW * 16) : Wt + (size_t)HP * KP;   // [4][32][XS]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float* Xs = Xs_all + wave * 32 * XS;
    GNM_LSTAMP(0)
    if constexpr (SPLITD) {
        for (int e = tid; e < EW; e += 256) {
            const int n = e & 31, kg = (e >> 5) & 1, c = (e >> 6) % KT, m = e / (64 * KT);
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                f[j] = (!NARROW || 32 * c + n < p.K) ? p.W[(size_t)(8 * m + 32 * kg + j) * p.ldw + 32 * c + n] : 0.f;
            u32x4 p1, p2, p3;
            lin_split8(f, p1, p2, p3);
            Wp[e] = p1; Wp[EW + e] = p2; Wp[2 * EW + e] = p3;
        }
    } else {
        for (int idx = tid; idx < HP * KP; idx += 256) {
            const int hh = idx / KP, k = idx - hh * KP;
            Wt[idx] = (!NARROW || k < p.K) ? p.W[(size_t)hh * p.ldw + k] : 0.f;
        }
    }
    __syncthreads();
    GNM_LSTAMP(1)
    int tk = 0;

    const int c4 = lane % H4, lrow0 = lane / H4;
    // BatchNorm-backward coefficients of this lane's column chunk: kept in registers for the whole kernel, except in
    // the SAMEZ variant, which needs those 20 registers across the MFMA phases and re-reads them (L1 hits) per tile
    float4 mu, rs, ca, a1, a2;
    if constexpr (!SAMEZ) {
        mu = *reinterpret_cast<const float4*>(p.mean + 4 * c4);
        rs = *reinterpret_cast<const float4*>(p.rstd + 4 * c4);
        ca = *reinterpret_cast<const float4*>(p.cA + 4 * c4);
        a1 = *reinterpret_cast<const float4*>(p.m1 + 4 * c4);
        a2 = *reinterpret_cast<const float4*>(p.m2 + 4 * c4);
    }
    float psc[KT], psh[KT];
#pragma unroll
    for (int b = 0; b < KT; ++b) {
        const bool kin = !NARROW || 32 * b + i < p.K;
        psc[b] = (p.pro_scale && kin) ? p.pro_scale[32 * b + i] : 1.f;
        psh[b] = (p.pro_scale && kin) ? p.pro_shift[32 * b + i] : 0.f;
    }
    f32x16 wacc[HT][KT];                          // dW accumulators
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) wacc[a][b][r] = 0.f;
    float dbacc[HT];
#pragma unroll
    for (int a = 0; a < HT; ++a) dbacc[a] = 0.f;
    // lower-BatchNorm statistics of the masked dX (this lane's 16-B column chunk oc4 of the output rows)
    constexpr int OR = 64 / O4;                   // output rows covered per pass of the store loop
    const int oc4 = lane % O4, orow0 = lane / O4;
    float4 ss1 = make_float4(0.f, 0.f, 0.f, 0.f), ss2 = ss1;
    float cmu[KT], crs[KT], cs1[KT], cs2[KT];     // SAMEZ: this lane's columns 32b + i of the lower BatchNorm
#pragma unroll
    for (int b = 0; b < KT; ++b) {
        cmu[b] = SAMEZ ? p.s_mean[32 * b + i] : 0.f;
        crs[b] = SAMEZ ? p.s_rstd[32 * b + i] : 0.f;
        cs1[b] = 0.f; cs2[b] = 0.f;
    }

    const int ntiles = (p.N + 31) / 32;
    const int nlast = p.N - 1;
    for (int t = lin_first_tile(wave, 1, gridDim.x); t < ntiles; t += gridDim.x * 4) {
        const int r0 = t * 32;
        GNM_LSTAMP(2 + 5 * min(tk, 11))
        // ---- all global loads of the tile first -------------------------------------
        float4 g4[NLD], z4[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int grow = min(r0 + lrow0 + j * RSTEP, nlast);
            g4[j] = *reinterpret_cast<const float4*>(p.G + (size_t)grow * p.ldg + 4 * c4);
#ifdef GNM_EXP_NOZ          // timing experiment only (wrong results): what the kernel costs without the Z stream
            z4[j] = g4[j];
#else
            z4[j] = *reinterpret_cast<const float4*>(p.Z + (size_t)grow * p.ldz + 4 * c4);
#endif
        }
        if constexpr (SAMEZ) {
            mu = *reinterpret_cast<const float4*>(p.mean + 4 * c4);
            rs = *reinterpret_cast<const float4*>(p.rstd + 4 * c4);
            ca = *reinterpret_cast<const float4*>(p.cA + 4 * c4);
            a1 = *reinterpret_cast<const float4*>(p.m1 + 4 * c4);
            a2 = *reinterpret_cast<const float4*>(p.m2 + 4 * c4);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- dZ tile -> LDS (rows past N are zero: they must not contribute) ----------
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int lrow = lrow0 + j * RSTEP;
            float4 d;
            d.x = ca.x * (g4[j].x - a1.x - (z4[j].x - mu.x) * rs.x * a2.x);
            d.y = ca.y * (g4[j].y - a1.y - (z4[j].y - mu.y) * rs.y * a2.y);
            d.z = ca.z * (g4[j].z - a1.z - (z4[j].z - mu.z) * rs.z * a2.z);
            d.w = ca.w * (g4[j].w - a1.w - (z4[j].w - mu.w) * rs.w * a2.w);
            if (r0 + lrow >= p.N) d = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(Xs + lrow * XS + 4 * c4) = d;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // f(X) operands of the wgrad product: issued now (the G/Z registers are dead), they are in
        // flight during the 64 dgrad MFMAs below
        float xv[16][KT];                         // X[r0 + 2s + h][32b + i]
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // contraction index of the wgrad product: any order, as long as dZ (below) uses the same one.  SAMEZ takes
            // the 32x32 accumulator's row order, so that xv[s] sits on the rows of dacc[.][s]
            const int krow = SAMEZ ? (s & 3) + 8 * (s >> 2) + 4 * h : 2 * s + h;
            const int grow = min(r0 + krow, nlast);
#pragma unroll
            for (int b = 0; b < KT; ++b)
                xv[s][b] = (!NARROW || 32 * b + i < p.K) ? p.X[(size_t)grow * p.ldx + 32 * b + i] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        GNM_LSTAMP(3 + 5 * min(tk, 11))
        // ---- dX = dZ W ------------------------------------------------------------------
        f32x16 dacc[KT];
        if constexpr (SPLITD) {
            if (p.dA) {
#pragma unroll
                for (int c = 0; c < KT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dacc[c][r] = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float4 v0 = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 8 * m);
                    const float4 v1 = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 8 * m + 4);
                    const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    u32x4 A1, A2, A3;
                    lin_split8(f, A1, A2, A3);
                    const lin_bf16x8 x1 = __builtin_bit_cast(lin_bf16x8, A1), x2 = __builtin_bit_cast(lin_bf16x8, A2),
                                     x3 = __builtin_bit_cast(lin_bf16x8, A3);
#pragma unroll
                    for (int c = 0; c < KT; ++c) {
                        const int e = (m * KT + c) * 64 + lane;
                        const lin_bf16x8 b1 = __builtin_bit_cast(lin_bf16x8, Wp[e]),
                                         b2 = __builtin_bit_cast(lin_bf16x8, Wp[EW + e]),
                                         b3 = __builtin_bit_cast(lin_bf16x8, Wp[2 * EW + e]);
                        dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b3, dacc[c], 0, 0, 0);
                        dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, b1, dacc[c], 0, 0, 0);
                        dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b2, dacc[c], 0, 0, 0);
                        dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b2, dacc[c], 0, 0, 0);
                        dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b1, dacc[c], 0, 0, 0);
                        dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b1, dacc[c], 0, 0, 0);
                    }
                }
            }
        } else if (p.dA) {
            float a[KH];
#pragma unroll
            for (int j = 0; j < KH / 4; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 4 * j);
                a[4 * j + 0] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
            }
#pragma unroll
            for (int c = 0; c < KT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[c][r] = 0.f;
            const float* wrow = Wt + (size_t)(KH * h) * KP + i;
#pragma unroll
            for (int s = 0; s < KH; ++s) {
#pragma unroll
                for (int c = 0; c < KT; ++c)
                    dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wrow[s * KP + 32 * c], dacc[c], 0, 0, 0);
            }
        }
        GNM_LSTAMP(4 + 5 * min(tk, 11))
        // ---- dW += dZ^T f(X), db += column sums of dZ -----------------------------------
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float av[HT];
            const int krow = SAMEZ ? (s & 3) + 8 * (s >> 2) + 4 * h : 2 * s + h;
#pragma unroll
            for (int a = 0; a < HT; ++a) av[a] = Xs[krow * XS + 32 * a + i];
            float xf[KT];
#pragma unroll
            for (int b = 0; b < KT; ++b) {
                float x = xv[s][b];
                if (p.pro_scale) {
                    x = x * psc[b] + psh[b];
                    if (p.pro_relu) x = gnm_relu(x);
                    if (NARROW && !(32 * b + i < p.K)) x = 0.f;      // padding columns stay zero
                }
                xf[b] = x;                                          // (SAMEZ keeps the raw Z in xv for the epilogue)
            }
#pragma unroll
            for (int a = 0; a < HT; ++a) {
                dbacc[a] += av[a];
#pragma unroll
                for (int b = 0; b < KT; ++b)
                    wacc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], xf[b], wacc[a][b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // everyone is done reading the dZ image
        GNM_LSTAMP(5 + 5 * min(tk, 11))
        // ---- store dX through the staging image (16-B row-contiguous stores) ---------------
        if (p.dA) {
            if constexpr (SAMEZ) {
                // ReLU mask of the lower BatchNorm and its backward sums, on the accumulator itself: lane (i, h) holds
                // column 32c + i of rows (r & 3) + 8 (r >> 2) + 4h, and xv[r][c] is Z of exactly that element
#pragma unroll
                for (int c = 0; c < KT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float z = xv[r][c];
                        float g = dacc[c][r];
                        if (!(z * psc[c] + psh[c] > 0.f)) g = 0.f;
                        if (r0 + (r & 3) + 8 * (r >> 2) + 4 * h < p.N) {
                            cs1[c] += g;
                            cs2[c] += g * ((z - cmu[c]) * crs[c]);
                        }
                        dacc[c][r] = g;
                    }
            }
#pragma unroll
            for (int c = 0; c < KT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) Xs[((r & 3) + 8 * (r >> 2) + 4 * h) * XS + 32 * c + i] = dacc[c][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if constexpr (STATS && !SAMEZ) {
                __builtin_amdgcn_sched_barrier(0);   // keep these loads below the MFMA phases (register peak)
                // coefficient vectors are (re)loaded here, not kept across the MFMA phases: they are
                // L1/L2 hits and 16 registers held for the whole tile would cost a wave of occupancy
                const float4 lsc = *reinterpret_cast<const float4*>(p.s_scale + 4 * oc4);
                const float4 lsh = *reinterpret_cast<const float4*>(p.s_shift + 4 * oc4);
                const float4 lmu = *reinterpret_cast<const float4*>(p.s_mean + 4 * oc4);
                const float4 lrs = *reinterpret_cast<const float4*>(p.s_rstd + 4 * oc4);
                constexpr int NJ = 32 / OR, HB = NJ > 4 ? 4 : NJ;     // 4 rows (16 registers of Z) at a time
#pragma unroll
                for (int j0 = 0; j0 < NJ; j0 += HB) {
                float4 zl[HB];
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const int grow = min(r0 + orow0 + (j0 + j) * OR, nlast);
                    zl[j] = *reinterpret_cast<const float4*>(p.sZ + (size_t)grow * p.ldsz + 4 * oc4);
                }
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const int row = orow0 + (j0 + j) * OR;
                    float4 g = *reinterpret_cast<const float4*>(Xs + row * XS + 4 * oc4);
                    const float4 z = zl[j];
                    if (!(z.x * lsc.x + lsh.x > 0.f)) g.x = 0.f;
                    if (!(z.y * lsc.y + lsh.y > 0.f)) g.y = 0.f;
                    if (!(z.z * lsc.z + lsh.z > 0.f)) g.z = 0.f;
                    if (!(z.w * lsc.w + lsh.w > 0.f)) g.w = 0.f;
                    if (r0 + row < p.N) {
                        *reinterpret_cast<float4*>(p.dA + (size_t)(r0 + row) * p.lda + 4 * oc4) = g;
                        ss1.x += g.x; ss1.y += g.y; ss1.z += g.z; ss1.w += g.w;
                        ss2.x += g.x * ((z.x - lmu.x) * lrs.x); ss2.y += g.y * ((z.y - lmu.y) * lrs.y);
                        ss2.z += g.z * ((z.z - lmu.z) * lrs.z); ss2.w += g.w * ((z.w - lmu.w) * lrs.w);
                    }
                }
                }
            } else if constexpr (NARROW) {
                for (int idx = lane; idx < 32 * p.K; idx += 64) {
                    const int row = idx / p.K, col = idx - row * p.K;
                    if (r0 + row < p.N) p.dA[(size_t)(r0 + row) * p.lda + col] = Xs[row * XS + col];
                }
            } else {
#pragma unroll
                for (int idx = lane; idx < 32 * O4; idx += 64) {
                    const int row = idx / O4, oc = idx - row * O4;
                    if (r0 + row < p.N)
                        *reinterpret_cast<float4*>(p.dA + (size_t)(r0 + row) * p.lda + 4 * oc) =
                            *reinterpret_cast<const float4*>(Xs + row * XS + 4 * oc);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        GNM_LSTAMP(6 + 5 * min(tk, 11))
        ++tk;
    }
    GNM_LSTAMP(62)

    // ---- lower-BatchNorm sums: lanes with the same column chunk, then the 4 waves (fixed order) ----
    if constexpr (STATS) {
        __syncthreads();
        double* sred = reinterpret_cast<double*>(smem);           // [4 waves][2][KP]
        if constexpr (SAMEZ) {
#pragma unroll
            for (int c = 0; c < KT; ++c) {                          // the two half-waves hold the same columns
                double d1 = (double)cs1[c], d2 = (double)cs2[c];
                d1 += __shfl_xor(d1, 32, 64);
                d2 += __shfl_xor(d2, 32, 64);
                if (h == 0) {
                    sred[(wave * 2 + 0) * KP + 32 * c + i] = d1;
                    sred[(wave * 2 + 1) * KP + 32 * c + i] = d2;
                }
            }
        }
        float v1[4] = {ss1.x, ss1.y, ss1.z, ss1.w}, v2[4] = {ss2.x, ss2.y, ss2.z, ss2.w};
#pragma unroll
        for (int c = 0; c < (SAMEZ ? 0 : 4); ++c) {
            double d1 = (double)v1[c], d2 = (double)v2[c];
#pragma unroll
            for (int off = O4; off < 64; off <<= 1) {
                d1 += __shfl_xor(d1, off, 64);
                d2 += __shfl_xor(d2, off, 64);
            }
            if (lane < O4) {
                sred[(wave * 2 + 0) * KP + 4 * oc4 + c] = d1;
                sred[(wave * 2 + 1) * KP + 4 * oc4 + c] = d2;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * KP; idx += 256) {
            const int which = idx / KP, col = idx - which * KP;
            double sum = 0.0;
            for (int w = 0; w < 4; ++w) sum += sred[(w * 2 + which) * KP + col];
            p.s_partial[((size_t)blockIdx.x * 2 + which) * p.K + col] = sum;
        }
    }

    // ---- combine the 4 waves' dW / db in a fixed order: one partial per block -------------
    __syncthreads();
    constexpr int TILE = 16 * 64;
    float* dump = reinterpret_cast<float*>(smem);                 // [4][HT*KT][16][64] + [4][HT][64]
    float* mine = dump + (size_t)wave * HT * KT * TILE;
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(a * KT + b) * TILE + r * 64 + lane] = wacc[a][b][r];
    float* dbdump = dump + (size_t)4 * HT * KT * TILE;
#pragma unroll
    for (int a = 0; a < HT; ++a) dbdump[(wave * HT + a) * 64 + lane] = dbacc[a];
    __syncthreads();
    float* out = p.partial + (size_t)blockIdx.x * ((size_t)p.H * p.K + p.H);
    for (int idx = tid; idx < HT * KT * TILE; idx += 256) {
        const int ab = idx / TILE;
        const int rl = idx - ab * TILE;
        const int r = rl >> 6, ln = rl & 63;
        const int a = ab / KT, b = ab - a * KT;
        const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int col = 32 * b + (ln & 31);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += dump[(size_t)w * HT * KT * TILE + idx];
        if (!NARROW || col < p.K) out[(size_t)row * p.K + col] = sum;
    }
    for (int idx = tid; idx < HT * 32; idx += 256) {
        const int a = idx >> 5, ii = idx & 31;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += dbdump[(w * HT + a) * 64 + ii] + dbdump[(w * HT + a) * 64 + 32 + ii];
        out[(size_t)p.H * p.K + 32 * a + ii] = sum;
    }
    GNM_LSTAMP(63)
}

// ---------------------------------------------------------------------------------
// The same fused backward for the form that has no statistics to reduce (the FIRST Linear of an MLP: its input is
// the aggregation output), software-pipelined across tiles: the G / Z tile of the wave's NEXT tile is requested in the
// middle of the current one -- behind the dgrad MFMAs, when the A fragments' 32 registers are free -- and travels
// under the wgrad MFMAs and the epilogue (the in-kernel timeline had a wave of the kernel above waiting for G and Z
// for 30-38 % of its life).  gfx950 retires vector-memory operations in issue order, so for that wait to be a counted
// one (not a drain that also waits for the dX stores issued after the request) every access between them is
// unconditional: tile loads and dX stores go through buffer descriptors that clip rows past N, and the first tile is
// peeled so that the loop is entered with the same queue its back edge carries (see gnm_lin_stream_kernel).
// The statistics variants above stay as they are: at 256 registers they have no room for a tile in flight (tried, in
// the shared template and as a SAMEZ flavour of this kernel with the request behind the wgrad MFMAs: 31-150 spills),
// and any restructuring of that kernel's body moved its register allocation.
// ---------------------------------------------------------------------------------
// SPLITD (K = H = 64): the dgrad product on the bf16 matrix pipe, dZ and W as three exact bf16 planes each (see
// gnm_lin_split_kernel).  wgrad keeps the fp32 instruction: on the bf16 pipe as well (a separate kernel, batch row as
// the contraction index, dZ columns from the LDS image and X rows from global split per tile) it was 12 % faster than
// the fp32 kernel stand-alone but spilled 18 dwords, and in the step it came out 0.3 % behind this form.
template <int KT, int HT, bool SPLITD = false>
__global__ void __launch_bounds__(256, 2) gnm_linear_bwd_pipe_kernel(const LbArgs p) {
    static_assert(!SPLITD || (KT == 2 && HT == 2), "SPLITD: K = H = 64 only");
    constexpr int KP = KT * 32, HP = HT * 32;
    constexpr int XS = (KP > HP ? KP : HP) + 4;
    constexpr int H4 = HP / 4;
    constexpr int NLD = (32 * H4) / 64;
    constexpr int RSTEP = 64 / H4;
    constexpr int KH = HP / 2;
    constexpr int O4 = KP / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EW = 4 * KT * 64;
    float* Wt = reinterpret_cast<float*>(smem);                   // [HP][KP]
    u32x4* Wp = reinterpret_cast<u32x4*>(smem);                   // SPLITD: [3][EW] bf16 operand entries of W instead
    float* Xs_all = SPLITD ? reinterpret_cast<float*>(smem + (size_t)3 * EW * 16) : Wt + (size_t)HP * KP;   // [4][32][XS]
    float* coef = Xs_all + 4 * 32 * XS;                           // [5][HP]: mean, rstd, cA, m1, m2 of the BatchNorm
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float* Xs = Xs_all + wave * 32 * XS;
    if constexpr (SPLITD) {
        for (int e = tid; e < EW; e += 256) {
            const int n = e & 31, kg = (e >> 5) & 1, c = (e >> 6) % KT, m = e / (64 * KT);
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = p.W[(size_t)(8 * m + 32 * kg + j) * p.ldw + 32 * c + n];
            u32x4 p1, p2, p3;
            lin_split8(f, p1, p2, p3);
            Wp[e] = p1; Wp[EW + e] = p2; Wp[2 * EW + e] = p3;
        }
    } else {
        for (int idx = tid; idx < HP * KP; idx += 256) {
            const int hh = idx / KP, k = idx - hh * KP;
            Wt[idx] = p.W[(size_t)hh * p.ldw + k];
        }
    }
    // the BatchNorm-backward coefficient vectors live in LDS and are read per tile (5 ds_read_b128: they count on
    // lgkmcnt, so re-reading them costs no place in the in-order vector-memory queue and no registers across the tile)
    for (int idx = tid; idx < HP; idx += 256) {
        coef[idx] = p.mean[idx]; coef[HP + idx] = p.rstd[idx]; coef[2 * HP + idx] = p.cA[idx];
        coef[3 * HP + idx] = p.m1[idx]; coef[4 * HP + idx] = p.m2[idx];
    }
    const int c4 = lane % H4, lrow0 = lane / H4;
    float psc[KT], psh[KT];
#pragma unroll
    for (int b = 0; b < KT; ++b) {
        psc[b] = p.pro_scale ? p.pro_scale[32 * b + i] : 1.f;
        psh[b] = p.pro_scale ? p.pro_shift[32 * b + i] : 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): nothing from the preamble is pending inside the tile loop
    __syncthreads();
    f32x16 wacc[HT][KT];
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) wacc[a][b][r] = 0.f;
    float dbacc[HT];
#pragma unroll
    for (int a = 0; a < HT; ++a) dbacc[a] = 0.f;

    const int ntiles = (p.N + 31) / 32;
    const int tstride = gridDim.x * 4;
    const int gz_voff_g = (lrow0 * p.ldg + 4 * c4) * 4, gz_step_g = RSTEP * p.ldg * 4;
    const int gz_voff_z = (lrow0 * p.ldz + 4 * c4) * 4, gz_step_z = RSTEP * p.ldz * 4;
    const int x_voff = (h * p.ldx + i) * 4;                       // X[r0 + 2 s + h][32 b + i]
    const int out_voff = ((lane / O4) * p.lda + 4 * (lane % O4)) * 4, out_step = (64 / O4) * p.lda * 4;
    // (no dX wanted: an empty descriptor over any readable address -- the stores stay unconditional)
    const float* dxbase = p.dA ? p.dA : p.G;

    u32x4 g4[NLD], z4[NLD];
    auto load_gz = [&](int tile) {
        const long long row0 = (long long)tile * 32;
        const long long rows = min((long long)p.N - row0, 32LL);
        const __amdgpu_buffer_rsrc_t rg = gnm_tile_rsrc(p.G + row0 * p.ldg, rows, p.ldg, HP);
        const __amdgpu_buffer_rsrc_t rz = gnm_tile_rsrc(p.Z + row0 * p.ldz, rows, p.ldz, HP);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            g4[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, gz_voff_g, j * gz_step_g, 0);
#ifdef GNM_EXP_NOZ
            z4[j] = g4[j];
#else
            z4[j] = __builtin_amdgcn_raw_buffer_load_b128(rz, gz_voff_z, j * gz_step_z, 0);
#endif
        }
    };
    auto do_tile = [&](int t, int t_next) {
        const int r0 = t * 32;
        const int rows = min(p.N - r0, 32);
        // ---- dZ tile -> LDS (rows past N are zero: they must not contribute) ----------
        const float4 mu = *reinterpret_cast<const float4*>(coef + 4 * c4);
        const float4 rs = *reinterpret_cast<const float4*>(coef + HP + 4 * c4);
        const float4 ca = *reinterpret_cast<const float4*>(coef + 2 * HP + 4 * c4);
        const float4 a1 = *reinterpret_cast<const float4*>(coef + 3 * HP + 4 * c4);
        const float4 a2 = *reinterpret_cast<const float4*>(coef + 4 * HP + 4 * c4);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int lrow = lrow0 + j * RSTEP;
            const float4 gj = __builtin_bit_cast(float4, g4[j]), zj = __builtin_bit_cast(float4, z4[j]);
            float4 d;
            d.x = ca.x * (gj.x - a1.x - (zj.x - mu.x) * rs.x * a2.x);
            d.y = ca.y * (gj.y - a1.y - (zj.y - mu.y) * rs.y * a2.y);
            d.z = ca.z * (gj.z - a1.z - (zj.z - mu.z) * rs.z * a2.z);
            d.w = ca.w * (gj.w - a1.w - (zj.w - mu.w) * rs.w * a2.w);
            if (lrow >= rows) d = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(Xs + lrow * XS + 4 * c4) = d;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // f(X) operands of the wgrad product: in flight during the dgrad MFMAs (clipped rows read 0; their dZ is 0)
        float xv[16][KT];
        {
            const __amdgpu_buffer_rsrc_t rx = gnm_tile_rsrc(p.X + (size_t)r0 * p.ldx, rows, p.ldx, KP);
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int b = 0; b < KT; ++b)
                    xv[s][b] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, x_voff, (2 * s * p.ldx + 32 * b) * 4, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- dX = dZ W ------------------------------------------------------------------
        f32x16 dacc[KT];
        if constexpr (SPLITD) {
#pragma unroll
            for (int c = 0; c < KT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[c][r] = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 v0 = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 8 * m);
                const float4 v1 = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 8 * m + 4);
                const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                u32x4 A1, A2, A3;
                lin_split8(f, A1, A2, A3);
                const lin_bf16x8 x1 = __builtin_bit_cast(lin_bf16x8, A1), x2 = __builtin_bit_cast(lin_bf16x8, A2),
                                 x3 = __builtin_bit_cast(lin_bf16x8, A3);
#pragma unroll
                for (int c = 0; c < KT; ++c) {
                    const int e = (m * KT + c) * 64 + lane;
                    const lin_bf16x8 b1 = __builtin_bit_cast(lin_bf16x8, Wp[e]), b2 = __builtin_bit_cast(lin_bf16x8, Wp[EW + e]),
                                     b3 = __builtin_bit_cast(lin_bf16x8, Wp[2 * EW + e]);
                    dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b3, dacc[c], 0, 0, 0);
                    dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, b1, dacc[c], 0, 0, 0);
                    dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b2, dacc[c], 0, 0, 0);
                    dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b2, dacc[c], 0, 0, 0);
                    dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b1, dacc[c], 0, 0, 0);
                    dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b1, dacc[c], 0, 0, 0);
                }
            }
        } else {
            float a[KH];
#pragma unroll
            for (int j = 0; j < KH / 4; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(Xs + i * XS + KH * h + 4 * j);
                a[4 * j + 0] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
            }
#pragma unroll
            for (int c = 0; c < KT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[c][r] = 0.f;
            const float* wrow = Wt + (size_t)(KH * h) * KP + i;
#pragma unroll
            for (int s = 0; s < KH; ++s) {
#pragma unroll
                for (int c = 0; c < KT; ++c)
                    dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wrow[s * KP + 32 * c], dacc[c], 0, 0, 0);
            }
        }
        load_gz(t_next);                          // past the wave's last tile: empty descriptors, no traffic
        // ---- dW += dZ^T f(X), db += column sums of dZ -----------------------------------
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float av[HT];
            const int krow = 2 * s + h;
#pragma unroll
            for (int a = 0; a < HT; ++a) av[a] = Xs[krow * XS + 32 * a + i];
            float xf[KT];
#pragma unroll
            for (int b = 0; b < KT; ++b) {
                float x = xv[s][b];
                if (p.pro_scale) {
                    x = x * psc[b] + psh[b];
                    if (p.pro_relu) x = gnm_relu(x);
                }
                xf[b] = x;
            }
#pragma unroll
            for (int a = 0; a < HT; ++a) {
                dbacc[a] += av[a];
#pragma unroll
                for (int b = 0; b < KT; ++b)
                    wacc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], xf[b], wacc[a][b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // everyone is done reading the dZ image
        // ---- store dX through the staging image (16-B row-contiguous, clipped buffer stores) ---------
#pragma unroll
        for (int c = 0; c < KT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) Xs[((r & 3) + 8 * (r >> 2) + 4 * h) * XS + 32 * c + i] = dacc[c][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const __amdgpu_buffer_rsrc_t rd = gnm_tile_rsrc(dxbase + (p.dA ? (size_t)r0 * p.lda : 0), p.dA ? rows : 0, p.lda, KP);
#pragma unroll
            for (int st = 0; st < (32 * O4) / 64; ++st) {
                const int idx = lane + 64 * st;
                const int row = idx / O4, oc = idx - row * O4;
                const u32x4 v = *reinterpret_cast<const u32x4*>(Xs + row * XS + 4 * oc);
                __builtin_amdgcn_raw_buffer_store_b128(v, rd, out_voff + st * out_step, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    {
        int t = lin_first_tile(wave, 1, gridDim.x);
        load_gz(t);
        if (t < ntiles) {
            do_tile(t, t + tstride);                  // peeled
            for (t += tstride; t < ntiles; t += tstride) do_tile(t, t + tstride);
        }
    }

    // ---- combine the 4 waves' dW / db in a fixed order: one partial per block (as in the kernel above) ----
    __syncthreads();
    constexpr int TILE = 16 * 64;
    float* dump = reinterpret_cast<float*>(smem);
    float* mine = dump + (size_t)wave * HT * KT * TILE;
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(a * KT + b) * TILE + r * 64 + lane] = wacc[a][b][r];
    float* dbdump = dump + (size_t)4 * HT * KT * TILE;
#pragma unroll
    for (int a = 0; a < HT; ++a) dbdump[(wave * HT + a) * 64 + lane] = dbacc[a];
    __syncthreads();
    float* out = p.partial + (size_t)blockIdx.x * ((size_t)p.H * p.K + p.H);
    for (int idx = tid; idx < HT * KT * TILE; idx += 256) {
        const int ab = idx / TILE;
        const int rl = idx - ab * TILE;
        const int r = rl >> 6, ln = rl & 63;
        const int a = ab / KT, b = ab - a * KT;
        const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int col = 32 * b + (ln & 31);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += dump[(size_t)w * HT * KT * TILE + idx];
        out[(size_t)row * p.K + col] = sum;
    }
    for (int idx = tid; idx < HT * 32; idx += 256) {
        const int a = idx >> 5, ii = idx & 31;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += dbdump[(w * HT + a) * 64 + ii] + dbdump[(w * HT + a) * 64 + 32 + ii];
        out[(size_t)p.H * p.K + 32 * a + ii] = sum;
    }
}

template <int KT, int HT, bool SPLITD = false>
static int launch_lb_pipe(const LbArgs& a, int grid, hipStream_t s) {
    constexpr int KP = KT * 32, HP = HT * 32;
    constexpr int XS = (KP > HP ? KP : HP) + 4;
    size_t lds = (SPLITD ? (size_t)3 * 4 * KT * 64 * 16 : (size_t)HP * KP * 4) + (size_t)4 * 32 * XS * 4 + (size_t)5 * HP * 4;
    const size_t dump = ((size_t)4 * HT * KT * 1024 + (size_t)4 * HT * 64) * 4;
    if (dump > lds) lds = dump;
    GNM_ALLOW_FULL_LDS((&gnm_linear_bwd_pipe_kernel<KT, HT, SPLITD>));
    hipLaunchKernelGGL((gnm_linear_bwd_pipe_kernel<KT, HT, SPLITD>), dim3(grid), dim3(256), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

template <int KT, int HT, bool STATS, bool NARROW = false, bool SAMEZ = false, bool SPLITD = false>
static int launch_lb(const LbArgs& a, int grid, hipStream_t s) {
    constexpr int KP = KT * 32, HP = HT * 32;
    constexpr int XS = (KP > HP ? KP : HP) + 4;
    size_t lds = (SPLITD ? (size_t)3 * 4 * KT * 64 * 16 : (size_t)HP * KP * 4) + (size_t)4 * 32 * XS * 4;
    const size_t dump = ((size_t)4 * HT * KT * 1024 + (size_t)4 * HT * 64) * 4;
    if (dump > lds) lds = dump;
    GNM_ALLOW_FULL_LDS((&gnm_linear_bwd_fused_kernel<KT, HT, STATS, NARROW, SAMEZ, SPLITD>));
    hipLaunchKernelGGL((gnm_linear_bwd_fused_kernel<KT, HT, STATS, NARROW, SAMEZ, SPLITD>), dim3(grid), dim3(256), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------
// The fused backward of a K = H = 64 Linear WITHOUT the Z stream (round 3).  The kernels above read four [N,64] arrays
// (G, Z, X in; dX out = 420 MB per launch of the headline batch) and Z -- this Linear's own output, needed only for
// xhat = (Z - mean) rstd inside the BatchNorm backward -- is a function of X, which the pass reads anyway for the weight
// gradient: Z = f(X) W^T + b is recomputed on the bf16 matrix pipe with exactly the instruction sequence of
// gnm_lin_split_kernel (48 MFMAs per 32 rows, the same values bit for bit) and a quarter of the traffic is gone.
// (Timing the old kernels with the Z loads stubbed out gave 96.5 -> 85.8 us and 92.5 -> 72.6 us: the bound of this form.)
// What changes around that:
//  * the recomputed Z arrives in the accumulator layout (lane = column, 16 rows in registers), so G is loaded in that
//    layout too (4-byte loads, 128 contiguous bytes per half-wave) and dZ is formed there; the BatchNorm coefficients
//    are per-lane scalars;
//  * the weight-gradient product takes its dZ operand from those registers (no column reads of the LDS image) and its
//    X operand from lane = column loads of the rows in the same order; WG16: both are split into three bf16 planes and
//    the product runs as 48 bf16 instructions instead of 64 fp32 ones (a quarter of the matrix-pipe time);
//  * only dgrad still needs the transpose: dZ goes through the wave's LDS image once, row-wise out;
//  * G AND the row-wise X fragments of the wave's next tile are requested before the weight-gradient product (the
//    registers Z used to occupy), in both forms -- the statistics form above had no room for that;
//  * eight waves per workgroup share the two weight images (forward orientation for Z, k-major for dgrad: 48 KB) and the
//    workgroup writes TWO rows of partials (waves 0-3, 4-7), so gnm_linear_bwd_grid(N) rows exist as before.
// SAMEZ: the second Linear of an MLP (the lower BatchNorm's input is X, its affine the prologue) -- ReLU mask and that
// BatchNorm's backward sums on the dX accumulators, as in gnm_linear_bwd_fused_kernel.  !SAMEZ: no lower statistics.
// ---------------------------------------------------------------------------------
static constexpr int kRzWaves = 8;

template <bool SAMEZ, bool WG16>
__global__ void __launch_bounds__(kRzWaves * 64) gnm_linear_bwd_rz_kernel(const LbArgs p) {
    constexpr int KP = 64, HP = 64, XS = 68, EW = 512, NW = kRzWaves, NT = NW * 64, TILE = 16 * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* Wf = reinterpret_cast<u32x4*>(smem);                   // [3][EW]: (m, c, lane = 32 kg + n) = W[32c+n][8m+32kg+j]
    u32x4* Wb = Wf + 3 * EW;                                      // [3][EW]: (m, c, lane = 32 kg + n) = W[8m+32kg+j][32c+n]
    float* Xs_all = reinterpret_cast<float*>(smem + (size_t)6 * EW * 16);     // [NW][32][XS]
    float* coef = Xs_all + NW * 32 * XS;                          // [6][64]: mean, rstd, cA, m1, m2, bias
    float* psv = coef + 6 * 64;                                   // [2][64]: prologue scale, shift
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float* Xs = Xs_all + wave * 32 * XS;
    const bool pro = p.pro_scale != nullptr;
    GNM_RSTAMP(0)
    for (int e = tid; e < EW; e += NT) {
        const int n = e & 31, kg = (e >> 5) & 1, c = (e >> 6) & 1, m = e >> 7;
        float f[8];
        u32x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = p.W[(size_t)(32 * c + n) * p.ldw + 8 * m + 32 * kg + j];
        lin_split8(f, p1, p2, p3);
        Wf[e] = p1; Wf[EW + e] = p2; Wf[2 * EW + e] = p3;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = p.W[(size_t)(8 * m + 32 * kg + j) * p.ldw + 32 * c + n];
        lin_split8(f, p1, p2, p3);
        Wb[e] = p1; Wb[EW + e] = p2; Wb[2 * EW + e] = p3;
    }
    for (int idx = tid; idx < 64; idx += NT) {
        coef[idx] = p.mean[idx]; coef[64 + idx] = p.rstd[idx]; coef[128 + idx] = p.cA[idx];
        coef[192 + idx] = p.m1[idx]; coef[256 + idx] = p.m2[idx]; coef[320 + idx] = p.bias ? p.bias[idx] : 0.f;
        psv[idx] = pro ? p.pro_scale[idx] : 1.f;
        psv[64 + idx] = pro ? p.pro_shift[idx] : 0.f;
    }
    // SAMEZ: the lower BatchNorm's mean / rstd and the lane's running sums live in LDS too (with them and the prologue
    // vectors in registers the statistics form spilled 126-191 registers; the plain form sits at 256 exactly)
    float psc[2], psh[2];                                         // (plain form: registers -- from LDS it spilled 34)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        psc[b] = (pro && !SAMEZ) ? p.pro_scale[32 * b + i] : 1.f;
        psh[b] = (pro && !SAMEZ) ? p.pro_shift[32 * b + i] : 0.f;
    }
    float* lstat = psv + 2 * 64;                                  // [2][64]: lower mean, rstd
    float* lsum = lstat + 2 * 64 + wave * 4 * 64;                 // [NW][2 c][2][64 lanes]: this wave's sums
    if constexpr (SAMEZ) {
        for (int idx = tid; idx < 64; idx += NT) {
            lstat[idx] = p.s_mean[idx];
            lstat[64 + idx] = p.s_rstd[idx];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) lsum[q * 64 + lane] = 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): nothing from the preamble is pending inside the tile loop
    __syncthreads();
    GNM_RSTAMP(1)
    int tk = 0;

    f32x16 wacc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) wacc[a][b][r] = 0.f;
    float dbacc[2] = {0.f, 0.f};

    const int gw = lin_first_tile(wave, NW / 4, p.part_rows);
    const int ntiles = (p.N + 31) / 32;
    const int tstride = p.part_rows * 4;
    const int g_voff = (4 * h * p.ldg + i) * 4;                   // G[row(r, h)][32 a + i], row(r, h) = (r&3) + 8(r>>2) + 4h
    const int xa_voff = (i * p.ldx + 32 * h) * 4;                 // X[i][32 h + 8 m + 0..7]: the forward's A fragments
    const int xc_voff = (4 * h * p.ldx + i) * 4;                  // X[row(r, h)][32 b + i]
    const int out_voff = ((lane >> 4) * p.lda + 4 * (lane & 15)) * 4, out_step = 4 * p.lda * 4;

    float g[2][16];
    u32x4 xa[4][2];
    auto load_next_x = [&](int tile) {
        const long long row0 = (long long)tile * 32;
        const long long rows = min((long long)p.N - row0, 32LL);
        const __amdgpu_buffer_rsrc_t rx = gnm_tile_rsrc(p.X + row0 * p.ldx, rows, p.ldx, KP);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            xa[m][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, xa_voff, 32 * m, 0);
            xa[m][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, xa_voff, 32 * m + 16, 0);
        }
    };
    auto load_next_g = [&](int tile) {
        const long long row0 = (long long)tile * 32;
        const long long rows = min((long long)p.N - row0, 32LL);
        const __amdgpu_buffer_rsrc_t rg = gnm_tile_rsrc(p.G + row0 * p.ldg, rows, p.ldg, HP);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                g[a][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, g_voff + 128 * a, ((r & 3) + 8 * (r >> 2)) * p.ldg * 4, 0));
    };
    auto do_tile = [&](int t, int t_next) {
        const int r0 = t * 32;
        const int rows = min(p.N - r0, 32);
        GNM_RSTAMP(2 + 6 * min(tk, 9))
        // ---- Z = f(X) W^T (+ bias below): gnm_lin_split_kernel's sequence ------------------------------------
        f32x16 dz[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[c][r] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 v0 = __builtin_bit_cast(float4, xa[m][0]), v1 = __builtin_bit_cast(float4, xa[m][1]);
            float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (pro) {
                const int k0 = 32 * h + 8 * m;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 sc = *reinterpret_cast<const float4*>(psv + k0 + 4 * q);
                    const float4 sh = *reinterpret_cast<const float4*>(psv + 64 + k0 + 4 * q);
                    f[4 * q + 0] = f[4 * q + 0] * sc.x + sh.x; f[4 * q + 1] = f[4 * q + 1] * sc.y + sh.y;
                    f[4 * q + 2] = f[4 * q + 2] * sc.z + sh.z; f[4 * q + 3] = f[4 * q + 3] * sc.w + sh.w;
                }
                if (p.pro_relu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = gnm_relu(f[j]);
                }
            }
            u32x4 A1, A2, A3;
            lin_split8(f, A1, A2, A3);
            const lin_bf16x8 a1 = __builtin_bit_cast(lin_bf16x8, A1), a2 = __builtin_bit_cast(lin_bf16x8, A2),
                             a3 = __builtin_bit_cast(lin_bf16x8, A3);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int e = (m * 2 + c) * 64 + lane;
                const lin_bf16x8 b1 = __builtin_bit_cast(lin_bf16x8, Wf[e]), b2 = __builtin_bit_cast(lin_bf16x8, Wf[EW + e]),
                                 b3 = __builtin_bit_cast(lin_bf16x8, Wf[2 * EW + e]);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, dz[c], 0, 0, 0);
            }
        }
        GNM_RSTAMP(3 + 6 * min(tk, 9))
        // ---- dZ = cA (G - m1 - xhat m2) on the accumulators; rows past N are zero; image for dgrad ------------
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int col = 32 * a + i;
            const float mu = coef[col], rs = coef[64 + col], ca = coef[128 + col], a1 = coef[192 + col],
                        a2 = coef[256 + col], bz = coef[320 + col];
            float dsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float z = dz[a][r] + bz;
                float d = ca * (g[a][r] - a1 - (z - mu) * rs * a2);
                if (lrow >= rows) d = 0.f;
                dz[a][r] = d;
                dsum += d;
                Xs[lrow * XS + col] = d;
            }
            dbacc[a] += dsum;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // X in the accumulators' row order (wgrad operand; SAMEZ: the values under the dX accumulators): in flight
        // during the dgrad MFMAs (cache hits: the tile was read row-wise for Z)
        float xv[16][2];
        {
            const __amdgpu_buffer_rsrc_t rx = gnm_tile_rsrc(p.X + (size_t)r0 * p.ldx, rows, p.ldx, KP);
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    xv[r][b] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xc_voff + 128 * b, ((r & 3) + 8 * (r >> 2)) * p.ldx * 4, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        GNM_RSTAMP(4 + 6 * min(tk, 9))
        // ---- dX = dZ W -------------------------------------------------------------------------------------------
        f32x16 dacc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[c][r] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 v0 = *reinterpret_cast<const float4*>(Xs + i * XS + 32 * h + 8 * m);
            const float4 v1 = *reinterpret_cast<const float4*>(Xs + i * XS + 32 * h + 8 * m + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            u32x4 A1, A2, A3;
            lin_split8(f, A1, A2, A3);
            const lin_bf16x8 x1 = __builtin_bit_cast(lin_bf16x8, A1), x2 = __builtin_bit_cast(lin_bf16x8, A2),
                             x3 = __builtin_bit_cast(lin_bf16x8, A3);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int e = (m * 2 + c) * 64 + lane;
                const lin_bf16x8 b1 = __builtin_bit_cast(lin_bf16x8, Wb[e]), b2 = __builtin_bit_cast(lin_bf16x8, Wb[EW + e]),
                                 b3 = __builtin_bit_cast(lin_bf16x8, Wb[2 * EW + e]);
                dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b3, dacc[c], 0, 0, 0);
                dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, b1, dacc[c], 0, 0, 0);
                dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b2, dacc[c], 0, 0, 0);
                dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b2, dacc[c], 0, 0, 0);
                dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b1, dacc[c], 0, 0, 0);
                dacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b1, dacc[c], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // the dZ image has been read
        GNM_RSTAMP(5 + 6 * min(tk, 9))
        // ---- dX out (SAMEZ: masked by the lower ReLU, and that BatchNorm's backward sums) ---------------------------
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // (read once per column block: between the image stores below the compiler cannot keep an LDS value)
            const float esc = SAMEZ ? psv[32 * c + i] : 1.f, esh = SAMEZ ? psv[64 + 32 * c + i] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                float gg = dacc[c][r];
                if constexpr (SAMEZ) {
                    const float z = xv[r][c];
                    if (!(z * esc + esh > 0.f)) gg = 0.f;
                    dacc[c][r] = gg;
                }
                Xs[lrow * XS + 32 * c + i] = gg;
            }
        }
        if constexpr (SAMEZ) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float lmu = lstat[32 * c + i], lrs = lstat[64 + 32 * c + i];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if ((r & 3) + 8 * (r >> 2) + 4 * h < rows) {
                        s1 += dacc[c][r];
                        s2 += dacc[c][r] * ((xv[r][c] - lmu) * lrs);
                    }
                }
                lsum[(2 * c + 0) * 64 + lane] += s1;
                lsum[(2 * c + 1) * 64 + lane] += s2;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const __amdgpu_buffer_rsrc_t rd = gnm_tile_rsrc(p.dA + (size_t)r0 * p.lda, rows, p.lda, KP);
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int idx = lane + 64 * st;
                const int row = idx >> 4, oc = idx & 15;
                const u32x4 v = *reinterpret_cast<const u32x4*>(Xs + row * XS + 4 * oc);
                __builtin_amdgcn_raw_buffer_store_b128(v, rd, out_voff + st * out_step, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GNM_RSTAMP(6 + 6 * min(tk, 9))
        // the next tile: G before the weight-gradient product; the row-wise X fragments too where the registers allow it
        // (SAMEZ carries eight more per-lane values through the tile and spilled 126 with both in flight: X follows the product)
        load_next_g(t_next);                      // past the wave's last tile: empty descriptors, no traffic
        if constexpr (!SAMEZ) load_next_x(t_next);
        // ---- dW += dZ^T f(X): both operands from registers, batch rows in the accumulators' order ---------------------
        if constexpr (WG16) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                lin_bf16x8 dp[2][3], xp[2][3];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float f[8];
                    u32x4 p1, p2, p3;
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = dz[a][8 * m + j];
                    lin_split8(f, p1, p2, p3);
                    dp[a][0] = __builtin_bit_cast(lin_bf16x8, p1); dp[a][1] = __builtin_bit_cast(lin_bf16x8, p2);
                    dp[a][2] = __builtin_bit_cast(lin_bf16x8, p3);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float x = xv[8 * m + j][a];
                        if (pro) {
                            x = SAMEZ ? x * psv[32 * a + i] + psv[64 + 32 * a + i] : x * psc[a] + psh[a];
                            if (p.pro_relu) x = gnm_relu(x);
                        }
                        f[j] = x;
                    }
                    lin_split8(f, p1, p2, p3);
                    xp[a][0] = __builtin_bit_cast(lin_bf16x8, p1); xp[a][1] = __builtin_bit_cast(lin_bf16x8, p2);
                    xp[a][2] = __builtin_bit_cast(lin_bf16x8, p3);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        wacc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][0], xp[b][2], wacc[a][b], 0, 0, 0);
                        wacc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][2], xp[b][0], wacc[a][b], 0, 0, 0);
                        wacc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][1], xp[b][1], wacc[a][b], 0, 0, 0);
                        wacc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][0], xp[b][1], wacc[a][b], 0, 0, 0);
                        wacc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][1], xp[b][0], wacc[a][b], 0, 0, 0);
                        wacc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[a][0], xp[b][0], wacc[a][b], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) {
                float xf[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float x = xv[sidx][b];
                    if (pro) {
                        x = SAMEZ ? x * psv[32 * b + i] + psv[64 + 32 * b + i] : x * psc[b] + psh[b];
                        if (p.pro_relu) x = gnm_relu(x);
                    }
                    xf[b] = x;
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        wacc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz[a][sidx], xf[b], wacc[a][b], 0, 0, 0);
            }
        }
        if constexpr (SAMEZ) {
            __builtin_amdgcn_sched_barrier(0);
            load_next_x(t_next);      // (requested before the product it cost 24 spilled registers per tile: 122 us)
        }
        GNM_RSTAMP(7 + 6 * min(tk, 9))
        ++tk;
    };
    {
        int t = gw;
        load_next_x(t);
        load_next_g(t);
        if (t < ntiles) {
            do_tile(t, t + tstride);                  // peeled (see gnm_linear_bwd_pipe_kernel)
            for (t += tstride; t < ntiles; t += tstride) do_tile(t, t + tstride);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    GNM_RSTAMP(62)

    // ---- lower-BatchNorm sums: two rows of partials per workgroup (waves 0-3, 4-7), fixed order ----
    if constexpr (SAMEZ) {
        __syncthreads();
        double* sred = reinterpret_cast<double*>(smem);           // [NW][2][KP]
        float cs1[2], cs2[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            cs1[c] = lsum[(2 * c + 0) * 64 + lane];
            cs2[c] = lsum[(2 * c + 1) * 64 + lane];
        }
        __syncthreads();                                          // (sred overlays nothing of lsum, but the weight images)
#pragma unroll
        for (int c = 0; c < 2; ++c) {                             // the two half-waves hold the same columns
            double d1 = (double)cs1[c], d2 = (double)cs2[c];
            d1 += __shfl_xor(d1, 32, 64);
            d2 += __shfl_xor(d2, 32, 64);
            if (h == 0) {
                sred[(wave * 2 + 0) * KP + 32 * c + i] = d1;
                sred[(wave * 2 + 1) * KP + 32 * c + i] = d2;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * 2 * KP; idx += NT) {
            const int grp = idx / (2 * KP), rest = idx - grp * 2 * KP;
            const int which = rest / KP, col = rest - which * KP;
            const int row = blockIdx.x * 2 + grp;
            if (row >= p.part_rows) continue;
            double sum = 0.0;
            for (int w = 4 * grp; w < 4 * grp + 4; ++w) sum += sred[(w * 2 + which) * KP + col];
            p.s_partial[((size_t)row * 2 + which) * p.K + col] = sum;
        }
    }
    // ---- dW / db: the waves of each half in a fixed order ----
    __syncthreads();
    float* dump = reinterpret_cast<float*>(smem);                 // [NW][4][16][64] + [NW][2][64]
    float* mine = dump + (size_t)wave * 4 * TILE;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(a * 2 + b) * TILE + r * 64 + lane] = wacc[a][b][r];
    float* dbdump = dump + (size_t)NW * 4 * TILE;
#pragma unroll
    for (int a = 0; a < 2; ++a) dbdump[(wave * 2 + a) * 64 + lane] = dbacc[a];
    __syncthreads();
    for (int idx = tid; idx < 2 * 4 * TILE; idx += NT) {
        const int grp = idx / (4 * TILE), rem = idx - grp * 4 * TILE;
        const int prow = blockIdx.x * 2 + grp;
        if (prow >= p.part_rows) continue;
        const int ab = rem / TILE;
        const int rl = rem - ab * TILE;
        const int r = rl >> 6, ln = rl & 63;
        const int a = ab >> 1, b = ab & 1;
        const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int col = 32 * b + (ln & 31);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += dump[(size_t)(4 * grp + w) * 4 * TILE + rem];
        p.partial[(size_t)prow * ((size_t)HP * KP + HP) + (size_t)row * KP + col] = sum;
    }
    for (int idx = tid; idx < 2 * 64; idx += NT) {
        const int grp = idx >> 6, a = (idx >> 5) & 1, ii = idx & 31;
        const int prow = blockIdx.x * 2 + grp;
        if (prow >= p.part_rows) continue;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += dbdump[((4 * grp + w) * 2 + a) * 64 + ii] + dbdump[((4 * grp + w) * 2 + a) * 64 + 32 + ii];
        p.partial[(size_t)prow * ((size_t)HP * KP + HP) + (size_t)HP * KP + 32 * a + ii] = sum;
    }
    GNM_RSTAMP(63)
}

template <bool SAMEZ, bool WG16>
static int launch_lb_rz(const LbArgs& a, int grid, hipStream_t s) {
    size_t lds = (size_t)6 * 512 * 16 + (size_t)kRzWaves * 32 * 68 * 4 + (size_t)(8 + 2 + kRzWaves * 4) * 64 * 4;
    const size_t dump = ((size_t)kRzWaves * 4 * 1024 + (size_t)kRzWaves * 2 * 64) * 4;
    if (dump > lds) lds = dump;
    GNM_ALLOW_FULL_LDS((&gnm_linear_bwd_rz_kernel<SAMEZ, WG16>));
    hipLaunchKernelGGL((gnm_linear_bwd_rz_kernel<SAMEZ, WG16>), dim3((grid + 1) / 2), dim3(kRzWaves * 64), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// ---------------------------------------------------------------------------------
// gnm_linear_bwd_rz_kernel for a NARROW input (K <= 16: the first Linear of the input layer, K = F0 = 7 in the benchmark),
// H = 64, no lower BatchNorm.  The Z-reading narrow form moves G and Z (210 MB at the headline batch) for an input that
// is 11 MB; here Z = X W^T + b is one 16-wide MFMA step per column tile, the weight gradient [64 x K] is one column tile
// (32 accumulators) and dX [32 x K] one: G is the only large stream.  Same structure as the K = 64 kernel (eight waves,
// two rows of partials per workgroup, G of the next tile requested before the weight-gradient product); X and dX rows
// are K floats wide and not 16-byte addressable: 4-byte accesses, clipped by their buffer descriptors.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(kRzWaves * 64) gnm_linear_bwd_rzn_kernel(const LbArgs p) {
    constexpr int HP = 64, XS = 68, NW = kRzWaves, NT = NW * 64, TILE = 16 * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* Wf = reinterpret_cast<u32x4*>(smem);                   // [3][2 c][64]: (c, lane = 32 kg + n) = W[32c+n][8kg+j], k < K
    u32x4* Wb = Wf + 3 * 128;                                     // [3][4 m][64]: (m, lane = 32 kg + n) = W[8m+32kg+j][n], n < K
    float* Xs_all = reinterpret_cast<float*>(smem + (size_t)(3 * 128 + 3 * 256) * 16);    // [NW][32][XS]
    float* coef = Xs_all + NW * 32 * XS;                          // [6][64]: mean, rstd, cA, m1, m2, bias
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    float* Xs = Xs_all + wave * 32 * XS;
    const int K = p.K;
    for (int e = tid; e < 128; e += NT) {
        const int n = e & 31, kg = (e >> 5) & 1, c = e >> 6;
        float f[8];
        u32x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 8 * kg + j < K ? p.W[(size_t)(32 * c + n) * p.ldw + 8 * kg + j] : 0.f;
        lin_split8(f, p1, p2, p3);
        Wf[e] = p1; Wf[128 + e] = p2; Wf[256 + e] = p3;
    }
    for (int e = tid; e < 256; e += NT) {
        const int n = e & 31, kg = (e >> 5) & 1, m = e >> 6;
        float f[8];
        u32x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = n < K ? p.W[(size_t)(8 * m + 32 * kg + j) * p.ldw + n] : 0.f;
        lin_split8(f, p1, p2, p3);
        Wb[e] = p1; Wb[256 + e] = p2; Wb[512 + e] = p3;
    }
    for (int idx = tid; idx < 64; idx += NT) {
        coef[idx] = p.mean[idx]; coef[64 + idx] = p.rstd[idx]; coef[128 + idx] = p.cA[idx];
        coef[192 + idx] = p.m1[idx]; coef[256 + idx] = p.m2[idx]; coef[320 + idx] = p.bias ? p.bias[idx] : 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    f32x16 wacc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[a][r] = 0.f;
    float dbacc[2] = {0.f, 0.f};
    const int gw = lin_first_tile(wave, NW / 4, p.part_rows);
    const int ntiles = (p.N + 31) / 32;
    const int tstride = p.part_rows * 4;
    const unsigned g_voff = (unsigned)((4 * h * p.ldg + i) * 4);
    const unsigned kclip = 0x80000000u;                           // past any buffer, no wrap with the row offsets
    const unsigned xc_voff = i < K ? (unsigned)((4 * h * p.ldx + i) * 4) : kclip;     // X[row(r, h)][i]
    const unsigned da_voff = i < K ? (unsigned)((4 * h * p.lda + i) * 4) : kclip;     // dX[row(r, h)][i]

    float g[2][16];
    auto load_next_g = [&](int tile) {
        const long long row0 = (long long)tile * 32;
        const long long rows = min((long long)p.N - row0, 32LL);
        const __amdgpu_buffer_rsrc_t rg = gnm_tile_rsrc(p.G + row0 * p.ldg, rows, p.ldg, HP);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                g[a][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, g_voff + 128 * a, ((r & 3) + 8 * (r >> 2)) * p.ldg * 4, 0));
    };
    auto do_tile = [&](int t, int t_next) {
        const int r0 = t * 32;
        const int rows = min(p.N - r0, 32);
        const __amdgpu_buffer_rsrc_t rx = gnm_tile_rsrc(p.X + (size_t)r0 * p.ldx, rows, p.ldx, K);
        // ---- Z = X W^T: one step of 16 (zero past K) -------------------------------------------------------------
        float fx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            fx[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rx, 8 * h + j < K ? (unsigned)((i * p.ldx + 8 * h + j) * 4) : kclip, 0, 0));
        float xv[16];                             // X in the accumulators' row order, column i: the wgrad operand
#pragma unroll
        for (int r = 0; r < 16; ++r)
            xv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xc_voff, ((r & 3) + 8 * (r >> 2)) * p.ldx * 4, 0));
        f32x16 dz[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[c][r] = 0.f;
        {
            u32x4 A1, A2, A3;
            lin_split8(fx, A1, A2, A3);
            const lin_bf16x8 a1 = __builtin_bit_cast(lin_bf16x8, A1), a2 = __builtin_bit_cast(lin_bf16x8, A2),
                             a3 = __builtin_bit_cast(lin_bf16x8, A3);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int e = c * 64 + lane;
                const lin_bf16x8 b1 = __builtin_bit_cast(lin_bf16x8, Wf[e]), b2 = __builtin_bit_cast(lin_bf16x8, Wf[128 + e]),
                                 b3 = __builtin_bit_cast(lin_bf16x8, Wf[256 + e]);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, dz[c], 0, 0, 0);
                dz[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, dz[c], 0, 0, 0);
            }
        }
        // ---- dZ = cA (G - m1 - xhat m2) on the accumulators; image for dgrad -----------------------------------------
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int col = 32 * a + i;
            const float mu = coef[col], rs = coef[64 + col], ca = coef[128 + col], a1 = coef[192 + col],
                        a2 = coef[256 + col], bz = coef[320 + col];
            float dsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float z = dz[a][r] + bz;
                float d = ca * (g[a][r] - a1 - (z - mu) * rs * a2);
                if (lrow >= rows) d = 0.f;
                dz[a][r] = d;
                dsum += d;
                Xs[lrow * XS + col] = d;
            }
            dbacc[a] += dsum;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- dX = dZ W: one column tile (columns past K are zero and not stored) ------------------------------------
        if (p.dA) {
            f32x16 dacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 v0 = *reinterpret_cast<const float4*>(Xs + i * XS + 32 * h + 8 * m);
                const float4 v1 = *reinterpret_cast<const float4*>(Xs + i * XS + 32 * h + 8 * m + 4);
                const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                u32x4 A1, A2, A3;
                lin_split8(f, A1, A2, A3);
                const lin_bf16x8 x1 = __builtin_bit_cast(lin_bf16x8, A1), x2 = __builtin_bit_cast(lin_bf16x8, A2),
                                 x3 = __builtin_bit_cast(lin_bf16x8, A3);
                const int e = m * 64 + lane;
                const lin_bf16x8 b1 = __builtin_bit_cast(lin_bf16x8, Wb[e]), b2 = __builtin_bit_cast(lin_bf16x8, Wb[256 + e]),
                                 b3 = __builtin_bit_cast(lin_bf16x8, Wb[512 + e]);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b3, dacc, 0, 0, 0);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, b1, dacc, 0, 0, 0);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b2, dacc, 0, 0, 0);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b2, dacc, 0, 0, 0);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, b1, dacc, 0, 0, 0);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, b1, dacc, 0, 0, 0);
            }
            const __amdgpu_buffer_rsrc_t rd = gnm_tile_rsrc(p.dA + (size_t)r0 * p.lda, rows, p.lda, K);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dacc[r]), rd, da_voff, ((r & 3) + 8 * (r >> 2)) * p.lda * 4, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // the dZ image has been read
        load_next_g(t_next);                      // past the wave's last tile: empty descriptors, no traffic
        // ---- dW += dZ^T X: both operands from registers, batch rows in the accumulators' order -----------------------
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float f[8];
            u32x4 p1, p2, p3;
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = xv[8 * m + j];
            lin_split8(f, p1, p2, p3);
            const lin_bf16x8 x1 = __builtin_bit_cast(lin_bf16x8, p1), x2 = __builtin_bit_cast(lin_bf16x8, p2),
                             x3 = __builtin_bit_cast(lin_bf16x8, p3);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = dz[a][8 * m + j];
                lin_split8(f, p1, p2, p3);
                const lin_bf16x8 d1 = __builtin_bit_cast(lin_bf16x8, p1), d2 = __builtin_bit_cast(lin_bf16x8, p2),
                                 d3 = __builtin_bit_cast(lin_bf16x8, p3);
                wacc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, x3, wacc[a], 0, 0, 0);
                wacc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d3, x1, wacc[a], 0, 0, 0);
                wacc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d2, x2, wacc[a], 0, 0, 0);
                wacc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, x2, wacc[a], 0, 0, 0);
                wacc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d2, x1, wacc[a], 0, 0, 0);
                wacc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, x1, wacc[a], 0, 0, 0);
            }
        }
    };
    {
        int t = gw;
        load_next_g(t);
        if (t < ntiles) {
            do_tile(t, t + tstride);
            for (t += tstride; t < ntiles; t += tstride) do_tile(t, t + tstride);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // ---- dW [64 x K] / db: the waves of each half in a fixed order ----
    __syncthreads();
    float* dump = reinterpret_cast<float*>(smem);                 // [NW][2][16][64] + [NW][2][64]
    float* mine = dump + (size_t)wave * 2 * TILE;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[a * TILE + r * 64 + lane] = wacc[a][r];
    float* dbdump = dump + (size_t)NW * 2 * TILE;
#pragma unroll
    for (int a = 0; a < 2; ++a) dbdump[(wave * 2 + a) * 64 + lane] = dbacc[a];
    __syncthreads();
    const size_t pstride = (size_t)HP * K + HP;
    for (int idx = tid; idx < 2 * 2 * TILE; idx += NT) {
        const int grp = idx / (2 * TILE), rem = idx - grp * 2 * TILE;
        const int prow = blockIdx.x * 2 + grp;
        if (prow >= p.part_rows) continue;
        const int a = rem / TILE;
        const int rl = rem - a * TILE;
        const int r = rl >> 6, ln = rl & 63;
        const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int col = ln & 31;
        if (col >= K) continue;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += dump[(size_t)(4 * grp + w) * 2 * TILE + rem];
        p.partial[(size_t)prow * pstride + (size_t)row * K + col] = sum;
    }
    for (int idx = tid; idx < 2 * 64; idx += NT) {
        const int grp = idx >> 6, a = (idx >> 5) & 1, ii = idx & 31;
        const int prow = blockIdx.x * 2 + grp;
        if (prow >= p.part_rows) continue;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += dbdump[((4 * grp + w) * 2 + a) * 64 + ii] + dbdump[((4 * grp + w) * 2 + a) * 64 + 32 + ii];
        p.partial[(size_t)prow * pstride + (size_t)HP * K + 32 * a + ii] = sum;
    }
}

static int launch_lb_rzn(const LbArgs& a, int grid, hipStream_t s) {
    size_t lds = (size_t)(3 * 128 + 3 * 256) * 16 + (size_t)kRzWaves * 32 * 68 * 4 + (size_t)6 * 64 * 4;
    const size_t dump = ((size_t)kRzWaves * 2 * 1024 + (size_t)kRzWaves * 2 * 64) * 4;
    if (dump > lds) lds = dump;
    GNM_ALLOW_FULL_LDS((&gnm_linear_bwd_rzn_kernel));
    hipLaunchKernelGGL(gnm_linear_bwd_rzn_kernel, dim3((grid + 1) / 2), dim3(kRzWaves * 64), lds, s, a);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

extern "C" int gnm_linear_bwd_grid(int N) {
    int g = ((N + 31) / 32 + 3) / 4;
    static const int cap = gnm_env_int("GNM_LINBWD_GRID", 512);   // tuning knob
    if (g > cap) g = cap;
    return g < 1 ? 1 : g;
}
extern "C" long long gnm_linear_bwd_workspace_floats(int N, int H, int K) {
    return (long long)gnm_linear_bwd_grid(N) * ((long long)H * K + H);
}

// Returns GNM_ERR_UNSUPPORTED (and launches nothing) when the shape/alignment is not
// eligible; the caller then uses gnm_bn_bwd_apply + gnm_linear_wgrad + gnm_linear_fwd.
extern "C" int gnm_linear_bwd_fused(const float* G, int ldg, const float* Z, int ldz, const float* mean,
                                    const float* rstd, const float* cA, const float* m1, const float* m2,
                                    const float* X, int ldx, const float* pro_scale, const float* pro_shift,
                                    int pro_relu, const float* W, int ldw, float* dA, int lda, float* dW, int lddw,
                                    float* db, float* workspace, int N, int K, int H, const float* sZ, int ldsz,
                                    const float* s_scale, const float* s_shift, const float* s_mean,
                                    const float* s_rstd, double* s_partial, void* stream) {
    if (N <= 0) return GNM_ERR_UNSUPPORTED;
    const bool narrow = K >= 1 && K < 32 && !sZ;            // zero-padded single K tile, scalar X / dX accesses
    if ((K != 32 && K != 64 && !narrow) || (H != 32 && H != 64)) return GNM_ERR_UNSUPPORTED;
    if ((ldg & 3) || (ldz & 3) || (dA && !narrow && (lda & 3)) || lin_force_generic()) return GNM_ERR_UNSUPPORTED;
    const uintptr_t al = reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Z) |
                         (narrow ? 0 : reinterpret_cast<uintptr_t>(dA)) | reinterpret_cast<uintptr_t>(mean) |
                         reinterpret_cast<uintptr_t>(rstd) | reinterpret_cast<uintptr_t>(cA) |
                         reinterpret_cast<uintptr_t>(m1) | reinterpret_cast<uintptr_t>(m2);
    if (al & 15) return GNM_ERR_UNSUPPORTED;
    if (sZ) {
        if (!dA || !s_partial || (ldsz & 3)) return GNM_ERR_BAD_ARG;
        const uintptr_t al2 = reinterpret_cast<uintptr_t>(sZ) | reinterpret_cast<uintptr_t>(s_scale) |
                              reinterpret_cast<uintptr_t>(s_shift) | reinterpret_cast<uintptr_t>(s_mean) |
                              reinterpret_cast<uintptr_t>(s_rstd);
        if (al2 & 15) return GNM_ERR_UNSUPPORTED;
    }
    LbArgs a;
    a.sZ = sZ; a.s_scale = s_scale; a.s_shift = s_shift; a.s_mean = s_mean; a.s_rstd = s_rstd;
    a.s_partial = s_partial; a.ldsz = ldsz;
    a.G = G; a.Z = Z; a.X = X; a.W = W; a.mean = mean; a.rstd = rstd; a.cA = cA; a.m1 = m1; a.m2 = m2;
    a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.dA = dA; a.partial = workspace;
    a.ldg = ldg; a.ldz = ldz; a.ldx = ldx; a.ldw = ldw; a.lda = lda; a.N = N; a.K = K; a.H = H; a.pro_relu = pro_relu;
    a.bias = nullptr; a.part_rows = 0;
#ifdef GNM_LIN_TUNING
    a.stamps = g_lin_stamps;
#else
    a.stamps = nullptr;
#endif
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = gnm_linear_bwd_grid(N);
    int rc = GNM_ERR_UNSUPPORTED;
    const int KT = narrow ? 0 : K / 32, HT = H / 32;
    // no statistics to reduce (the first Linear of an MLP): the cross-tile pipelined kernel
    const bool pipe = !narrow && !sZ && !linbwd_no_pipe();
    if (pipe) {
        if (KT == 1 && HT == 1) rc = launch_lb_pipe<1, 1>(a, grid, s);
        if (KT == 2 && HT == 1) rc = launch_lb_pipe<2, 1>(a, grid, s);
        if (KT == 1 && HT == 2) rc = launch_lb_pipe<1, 2>(a, grid, s);
        if (KT == 2 && HT == 2) rc = linbwd_no_split() ? launch_lb_pipe<2, 2>(a, grid, s) : launch_lb_pipe<2, 2, true>(a, grid, s);
    }
    if (narrow && HT == 1) rc = launch_lb<1, 1, false, true>(a, grid, s);
    if (narrow && HT == 2)
        rc = linbwd_no_split() ? launch_lb<1, 2, false, true>(a, grid, s) : launch_lb<1, 2, false, true, false, true>(a, grid, s);
    if (!pipe && KT == 1 && HT == 1) rc = sZ ? launch_lb<1, 1, true>(a, grid, s) : launch_lb<1, 1, false>(a, grid, s);
    if (!pipe && KT == 2 && HT == 1) rc = sZ ? launch_lb<2, 1, true>(a, grid, s) : launch_lb<2, 1, false>(a, grid, s);
    if (!pipe && KT == 1 && HT == 2) rc = sZ ? launch_lb<1, 2, true>(a, grid, s) : launch_lb<1, 2, false>(a, grid, s);
    // the second Linear of an MLP: the lower BatchNorm's input is this Linear's input and its affine is the prologue
    const bool samez = sZ && sZ == X && ldsz == ldx && s_scale == pro_scale && s_shift == pro_shift && pro_relu &&
                       !linbwd_no_samez();
    if (!pipe && KT == 2 && HT == 2)
        rc = samez ? (linbwd_no_split() ? launch_lb<2, 2, true, false, true>(a, grid, s)
                                        : launch_lb<2, 2, true, false, true, true>(a, grid, s))
                   : (sZ ? launch_lb<2, 2, true>(a, grid, s) : launch_lb<2, 2, false>(a, grid, s));
    if (rc != GNM_OK) return rc;
    if (!dW) return GNM_OK;      // deferred: the partials stay in `workspace` for gnm_reduce_partials_multi
    const long long stride = (long long)H * K + H;
    const int count = H * K + H;
    hipLaunchKernelGGL(gnm_reduce_partials_kernel, dim3((count + 31) / 32), dim3(1024), 0, s, workspace, grid, stride,
                       H, K, 0, dW, lddw, db);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// gnm_linear_bwd_fused for a Linear whose output Z the caller does NOT pass: Z = f(X) W^T + bias is recomputed
// (gnm_linear_bwd_rz_kernel).  K = H = 64 with dA wanted, and either no lower BatchNorm (sZ = NULL) or the lower
// BatchNorm's input being X with the prologue as its affine (the two Linears of the headline model's MLPs); anything
// else returns GNM_ERR_UNSUPPORTED and the caller takes gnm_linear_bwd_fused with the stored Z.  Workspace, partial
// rows and the deferred reduction (dW = NULL) are those of gnm_linear_bwd_fused.
extern "C" int gnm_linear_bwd_fused_rz(const float* G, int ldg, const float* bias, const float* mean, const float* rstd,
                                       const float* cA, const float* m1, const float* m2, const float* X, int ldx,
                                       const float* pro_scale, const float* pro_shift, int pro_relu, const float* W,
                                       int ldw, float* dA, int lda, float* dW, int lddw, float* db, float* workspace,
                                       int N, int K, int H, const float* sZ, int ldsz, const float* s_scale,
                                       const float* s_shift, const float* s_mean, const float* s_rstd,
                                       double* s_partial, void* stream) {
    static const bool off = gnm_env_int("GNM_LINBWD_NO_RZ", 0) != 0;
    static const bool wg16 = gnm_env_int("GNM_LINBWD_WG16", 1) != 0;     // A/B knob: weight gradient on the bf16 pipe
    const bool narrow = K >= 1 && K <= 16 && !sZ && !pro_scale;    // the input layer's first Linear (one 16-wide step of Z): gnm_linear_bwd_rzn_kernel
    if (off || N <= 0 || (K != 64 && !narrow) || H != 64 || (!dA && !narrow) || lin_force_generic() || linbwd_no_split())
        return GNM_ERR_UNSUPPORTED;
    if (!narrow && ((ldx & 3) || (lda & 3))) return GNM_ERR_UNSUPPORTED;
    if (!narrow && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dA)) & 15)) return GNM_ERR_UNSUPPORTED;
    if ((long long)32 * (ldg > ldx ? (ldg > lda ? ldg : lda) : (ldx > lda ? ldx : lda)) * 4 >= (1LL << 31)) return GNM_ERR_UNSUPPORTED;
    const bool samez = sZ != nullptr;
    // the statistics form is correct and tested but measured BEHIND the kernel that reads Z (101.9 vs 99.3 us at the
    // headline shape: its extra per-lane state lives in LDS); the plain form is ahead (87.9 vs 90.9 us).  GNM_LINBWD_RZ_STATS=1
    // enables it for A/B timing and for its test.
    static const bool rz_stats = gnm_env_int("GNM_LINBWD_RZ_STATS", 0) != 0;
    if (samez && !rz_stats) return GNM_ERR_UNSUPPORTED;
    if (samez && !(sZ == X && ldsz == ldx && s_scale == pro_scale && s_shift == pro_shift && pro_scale && pro_relu &&
                   s_partial && s_mean && s_rstd))
        return GNM_ERR_UNSUPPORTED;
    LbArgs a;
    a.sZ = sZ; a.s_scale = s_scale; a.s_shift = s_shift; a.s_mean = s_mean; a.s_rstd = s_rstd;
    a.s_partial = s_partial; a.ldsz = ldsz;
    a.G = G; a.Z = nullptr; a.X = X; a.W = W; a.mean = mean; a.rstd = rstd; a.cA = cA; a.m1 = m1; a.m2 = m2;
    a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.dA = dA; a.partial = workspace;
    a.ldg = ldg; a.ldz = 0; a.ldx = ldx; a.ldw = ldw; a.lda = lda; a.N = N; a.K = K; a.H = H; a.pro_relu = pro_relu;
#ifdef GNM_LIN_TUNING
    a.stamps = g_lin_stamps;
#else
    a.stamps = nullptr;
#endif
    a.bias = bias;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = gnm_linear_bwd_grid(N);
    a.part_rows = grid;
    int rc;
    if (narrow) rc = launch_lb_rzn(a, grid, s);
    else if (samez) rc = wg16 ? launch_lb_rz<true, true>(a, grid, s) : launch_lb_rz<true, false>(a, grid, s);
    else rc = wg16 ? launch_lb_rz<false, true>(a, grid, s) : launch_lb_rz<false, false>(a, grid, s);
    if (rc != GNM_OK) return rc;
    if (!dW) return GNM_OK;
    const long long stride = (long long)H * K + H;
    const int count = H * K + H;
    hipLaunchKernelGGL(gnm_reduce_partials_kernel, dim3((count + 31) / 32), dim3(1024), 0, s, workspace, grid, stride,
                       H, K, 0, dW, lddw, db);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// The dW / db partial reductions of several gnm_linear_bwd_fused calls (dW = NULL there) in ONE launch: nothing in
// the backward depends on a weight gradient, so the ten ~5-us reductions of a step need not sit between its kernels.
// Same summation order per element as gnm_reduce_partials_kernel.
#define GNM_MAX_REDUCE_JOBS 32
struct ReduceJobs {
    const float* partial[GNM_MAX_REDUCE_JOBS];
    float* dW[GNM_MAX_REDUCE_JOBS];
    float* db[GNM_MAX_REDUCE_JOBS];
    int nblk[GNM_MAX_REDUCE_JOBS], H[GNM_MAX_REDUCE_JOBS], K[GNM_MAX_REDUCE_JOBS], ldw[GNM_MAX_REDUCE_JOBS];
    int first_block[GNM_MAX_REDUCE_JOBS + 1];
    int njobs;
};
__global__ void __launch_bounds__(1024) gnm_reduce_partials_multi_kernel(const ReduceJobs J) {
    __shared__ float red[32][32];
    int j = 0;
    while (j + 1 < J.njobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;      // workgroup-uniform
    const float* __restrict__ partial = J.partial[j];
    const int nblk = J.nblk[j], H = J.H[j], kw = J.K[j];
    const long long stride = (long long)H * kw + H;
    const int tid = threadIdx.x;
    const int grp = tid >> 5, ngrp = (int)blockDim.x >> 5;
    const int e = ((int)blockIdx.x - J.first_block[j]) * 32 + (tid & 31);
    const int count = H * kw + H;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < count) {
        int b = grp;
        for (; b + 3 * ngrp < nblk; b += 4 * ngrp) {
            const float v0 = partial[(size_t)(b + 0 * ngrp) * stride + e], v1 = partial[(size_t)(b + 1 * ngrp) * stride + e];
            const float v2 = partial[(size_t)(b + 2 * ngrp) * stride + e], v3 = partial[(size_t)(b + 3 * ngrp) * stride + e];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; b < nblk; b += ngrp) s0 += partial[(size_t)b * stride + e];
    }
    red[grp][tid & 31] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (tid < 32 && e < count) {
        float s = 0.f;
        for (int g = 0; g < ngrp; ++g) s += red[g][tid];
        if (e < H * kw) {
            const int row = e / kw, col = e - row * kw;
            J.dW[j][(size_t)row * J.ldw[j] + col] = s;
        } else if (J.db[j]) {
            J.db[j][e - H * kw] = s;
        }
    }
}

// workspaces[i]: the workspace of the i-th gnm_linear_bwd_fused(N, K = Ks[i], H = Hs[i], dW = NULL) call of the same N.
extern "C" int gnm_reduce_partials_multi(const float* const* workspaces_host, float* const* dW_host, const int* lddw_host,
                                         float* const* db_host, const int* Hs_host, const int* Ks_host, int njobs, int N,
                                         void* stream) {
    if (njobs <= 0) return GNM_OK;
    if (njobs > GNM_MAX_REDUCE_JOBS || N <= 0) return GNM_ERR_BAD_ARG;
    ReduceJobs J;
    int blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        if (!workspaces_host[i] || !dW_host[i] || Hs_host[i] <= 0 || Ks_host[i] <= 0) return GNM_ERR_BAD_ARG;
        J.partial[i] = workspaces_host[i]; J.dW[i] = dW_host[i]; J.db[i] = db_host[i];
        J.nblk[i] = gnm_linear_bwd_grid(N); J.H[i] = Hs_host[i]; J.K[i] = Ks_host[i]; J.ldw[i] = lddw_host[i];
        J.first_block[i] = blocks;
        blocks += (Hs_host[i] * Ks_host[i] + Hs_host[i] + 31) / 32;
    }
    J.first_block[njobs] = blocks;
    J.njobs = njobs;
    hipLaunchKernelGGL(gnm_reduce_partials_multi_kernel, dim3(blocks), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), J);
    GNM_CHECK_LAUNCH();
    return GNM_OK;
}

// dW[H,K] (row-major, ld = ldw) and db[H] from dZ[N,H] and f(X)[N,K].  `workspace`
// must hold gnm_wgrad_workspace_floats(N,H,K) floats.
extern "C" int gnm_linear_wgrad(const float* dZ, int ldd, const float* X, int ldx, int N, int H, int K,
                                const float* pro_scale, const float* pro_shift, int pro_relu, float* dW, int ldw,
                                float* db, float* workspace, void* stream) {
    if (H <= 0 || K <= 0 || H > 128) return GNM_ERR_BAD_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = gnm_wgrad_grid(N > 0 ? N : 1);
    int rpb = ((N > 0 ? N : 1) + grid - 1) / grid;
    rpb = (rpb + 1) & ~1;
    const int HT = (H + 31) / 32;
    for (int k0 = 0; k0 < K; k0 += 128) {
        const int kw = (K - k0) < 128 ? (K - k0) : 128;
        const int KT = (kw + 31) / 32;
        WgArgs a;
        a.dZ = dZ; a.X = X; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.partial = workspace;
        a.ldd = ldd; a.ldx = ldx; a.N = N; a.H = H; a.K = K; a.k0 = k0; a.kw = kw; a.rows_per_block = rpb;
        a.pro_relu = pro_relu;
        const int WI = HT < 2 ? HT : 2, WJ = KT < 2 ? KT : 2;
        const int QI = (HT + WI - 1) / WI, QJ = (KT + WJ - 1) / WJ;
        int rc = GNM_ERR_UNSUPPORTED;
        const bool fast = (H % 32) == 0 && N > 0 && !lin_force_generic();
        // H = 128 and a full 128-column window, 32-bit offsets inside a workgroup's row range: the bf16-pipe kernel
        if (fast && H == 128 && kw == 128 && !lin_no_split() && (long long)(rpb + 64) * (ldd > ldx ? ldd : ldx) * 4 < (1LL << 31)) {
            rc = launch_wgrad_split128(a, grid, s);
            if (rc != GNM_OK) return rc;
        } else {
#define GNM_WG_CASE(WI_, WJ_, QI_, QJ_)                                                       \
    if (WI == WI_ && WJ == WJ_ && QI == QI_ && QJ == QJ_)                                     \
        rc = fast ? launch_wgrad_fast<WI_, WJ_, QI_, QJ_>(a, grid, s) : launch_wgrad<WI_, WJ_, QI_, QJ_>(a, grid, s);
        GNM_WG_CASE(1, 1, 1, 1) GNM_WG_CASE(1, 2, 1, 1) GNM_WG_CASE(1, 2, 1, 2)
        GNM_WG_CASE(2, 1, 1, 1) GNM_WG_CASE(2, 2, 1, 1) GNM_WG_CASE(2, 2, 1, 2)
        GNM_WG_CASE(2, 1, 2, 1) GNM_WG_CASE(2, 2, 2, 1) GNM_WG_CASE(2, 2, 2, 2)
#undef GNM_WG_CASE
        }
        if (rc != GNM_OK) return rc;
        const long long stride = (long long)H * kw + H;
        const int count = H * kw + H;
        hipLaunchKernelGGL(gnm_reduce_partials_kernel, dim3((count + 31) / 32), dim3(1024), 0, s, workspace, grid,
                           stride, H, kw, k0, dW, ldw, db);
        GNM_CHECK_LAUNCH();
    }
    return GNM_OK;
}
