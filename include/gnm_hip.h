/* gnm_hip.h -- C-ABI of libgnm_hip.so: the MI355X (gfx950) GIN message-passing hot path.
 *
 * Drop-in boundary for the hot path of egyptdj/graph-neural-mapping
 * (GIN_InfoMaxReg.forward / backward, /root/reference models/graphcnn.py:194-251).
 * The reference is pure Python on PyTorch; it has no FFI of its own, so each entry
 * point below names the reference call site (file:line) whose ATen work it replaces.
 * The Python host mirror (graph-neural-mapping_amd/models/graphcnn.py) binds these
 * through ctypes; see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - matrices are fp32 row-major with an explicit leading dimension (in floats);
 *   - `stream` is a hipStream_t passed as void*; nothing synchronises or allocates
 *     (hipGraph-capturable).  The only process state is launch configuration: a
 *     per-(kernel, device) flag that hipFuncSetAttribute(MaxDynamicSharedMemorySize) has been
 *     applied on that device, and tuning knobs read from the environment once per process;
 *   - return value: 0 on success, >0 a hipError_t, <0 GNM_ERR_* below.
 *
 * Batch description (replaces the int64 block-diagonal COO built per forward at
 * graphcnn.py:84-106 and the [B,N] readout COO at :109-134):
 *   rowptr / col      per-graph CSR arena: int32 row offsets (n_g + 1 per graph,
 *                     graph-local, starting at 0) and uint16 graph-local column ids;
 *   b_rp_off[b]       offset (elements) of batch graph b's rowptr block in `rowptr`;
 *   b_col_off[b]      offset of its column block in `col`; `col` must be 4-byte aligned and stay
 *                     readable for 256 ids past the end of the last block (the gather requests
 *                     two 128-id windows per row as aligned id pairs, unconditionally, and
 *                     discards what lies beyond the row's end);
 *   node_off[B+1]     first node row of each batch graph in the concatenated [N, F]
 *                     feature matrix (cumsum of len(graph.g), graphcnn.py:88-90).
 */
#ifndef GNM_HIP_H
#define GNM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNM_OK 0
#define GNM_ERR_BAD_ARG (-1)
#define GNM_ERR_UNSUPPORTED (-2)

const char* gnm_version(void);
/* Test hook: one instance of the per-device "configure once" guard that every launcher keeps for
 * hipFuncSetAttribute(MaxDynamicSharedMemorySize) -- 1 the first time `device` is seen, 0 afterwards;
 * reset != 0 clears it.  Touches no device. */
int gnm_debug_device_once(int device, int reset);

/* ---- host helpers (no GPU needed) ------------------------------------------------ */

/* CSR of one graph from the reference's edge_mat ([2,E] int64 host array, row-major:
 * sources then targets; util.py:99-103).  Stable in edge order, duplicates kept.
 * Replaces the per-forward COO assembly at graphcnn.py:89-93. */
int gnm_csr_from_edge_mat(const int64_t* edge_mat_host, long long E, int n, int32_t* rowptr_host,
                          uint16_t* col_host);
int gnm_csr_transpose(const int32_t* rowptr_host, const uint16_t* col_host, int n, int32_t* rowptr_t_host,
                      uint16_t* col_t_host);
int gnm_csr_is_symmetric(const int32_t* rowptr_host, const uint16_t* col_host, int n);
/* Order each CSR row's ids for the 32-float-slice gather (gnm_agg with F = 128, n ~ 1000: BASELINE configs[3]): position j
 * of a row gets an even id when (j & 3) < 2 and an odd one otherwise while the row has both kinds (relative order inside a
 * parity class kept).  Two 128-byte LDS rows of equal parity share their banks, and the hardware serves a ds_read_b128 in
 * lane groups that pair positions (8s, 8s+3), (8s+1, 8s+2), (8s+4, 8s+7), (8s+5, 8s+6).  The edge multiset -- all that
 * Adj_block (graphcnn.py:91-104) fixes -- is unchanged; every other kernel is indifferent to the order.  In place. */
int gnm_csr_parity_order(const int32_t* rowptr_host, uint16_t* col_host, int n);
/* Inverse of the above for a whole batch: the reference's Adj_block._indices()
 * (graphcnn.py:91-104), rows grouped by (graph, row).  Used by the parity tests. */
long long gnm_batch_coo_from_csr(const int32_t* rowptr_arena_host, const uint16_t* col_arena_host,
                                 const int64_t* b_rp_off_host, const int64_t* b_col_off_host,
                                 const int32_t* node_off_host, int B, int self_loops, int64_t* out_rows_host,
                                 int64_t* out_cols_host);

/* ---- neighbour aggregation --------------------------------------------------------
 * forward  (backward = 0): y = A x [/ deg] + (1 + eps) x      learn_eps  (graphcnn.py:154-161)
 *                          y = (A + I) x [/ (deg + 1)]        otherwise  (graphcnn.py:178-182, :97-102)
 * backward (backward = 1): the autograd transpose of the above over the TRANSPOSED CSR
 *                          (pass the forward CSR as deg_rowptr/b_deg_off; they may alias
 *                          rowptr/b_rp_off for symmetric graphs), plus, when deps_partial
 *                          is non-null, fp64 partials of d eps = sum(x * hfwd).
 * `eps` points at eps[layer] on the device (null: coefficient 1).  `self_loop` = !learn_eps.
 * n_max = largest graph of the batch; nnz_max = most edges of any batch graph (0 = unknown: narrow
 * feature slices then keep their column ids in global memory instead of LDS).
 * Replaces torch.spmm at graphcnn.py:154,157,178,181. */
int gnm_agg(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off, const int64_t* b_col_off,
            const int32_t* deg_rowptr, const int64_t* b_deg_off, const int32_t* node_off, int B, int n_max,
            int nnz_max, const float* x, int ldx, float* y, int ldy, int F, const float* eps, int average, int self_loop,
            int backward, const float* hfwd, int ldh, double* deps_partial, void* stream);
/* gnm_agg(backward = 1) for the layer-output gradient, fused with what gnm_bn_relu_bwd_stats would then do
 * for the BatchNorm+ReLU that produced that layer output (graphcnn.py:163-166 of the layer below): y becomes
 * G = (A^T x [+ self terms] + w_b*dpool[b] + dsc1[v]*U[b] + quirk rows) * relu-mask(sZ), and s_partial receives
 * [B][2][F] doubles (sum G, sum G*xhat) for gnm_bn_bwd_finalize.  Two shapes: F = 64 with the whole [n, 64] tile
 * in LDS, or an F that is whole 32-float slices where gnm_agg_slice_width(F, n_max) = 32 (hidden_dim 128 on graphs of
 * 625-1231 nodes, round 4); GNM_ERR_UNSUPPORTED otherwise (nothing is launched).  dpool / dsc1 (with U, inv_perm,
 * s2sum) may be null.
 * deps_partial with hfwd == NULL (learn_eps form only): the layer input h = relu(sZ*s_scale+s_shift) is recomputed in
 * the epilogue from the sZ row it already holds, so d eps costs no pass over h. */
int gnm_agg_bwd_stats(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off, const int64_t* b_col_off,
                      const int32_t* deg_rowptr, const int64_t* b_deg_off, const int32_t* node_off, int B, int n_max,
                      int nnz_max, const float* x, int ldx, float* y, int ldy, int F, const float* eps, int average,
                      int self_loop, const float* hfwd, int ldh, double* deps_partial, const float* sZ, int ldsz,
                      const float* s_scale, const float* s_shift, const float* s_mean, const float* s_rstd,
                      const float* dpool, int ld_dpool, int graph_avg, const float* dsc1, const float* U, int ld_U,
                      const int32_t* inv_perm, const float* s2sum, double* s_partial, void* stream);
/* gnm_agg of layer l+1 whose tile load IS layer l's outer BatchNorm + ReLU + graph readout
 * (graphcnn.py:163-166 and :229 folded into :154-161): z = output of layer l's last Linear, hout <- h_l =
 * relu(z*scale+shift) (hout may be NULL: the activation is then not written -- its other consumer, the
 * discriminator, can re-form it from z, see gnm_disc_score_fwd), gf[b,:] <- sum (mean when graph_avg) of h_l over
 * graph b (gf may be NULL), y <- the
 * aggregation of h_l.  Two shapes: F = 64 as one 64-wide LDS slice, or an F that is whole 32-float slices where
 * gnm_agg_slice_width(F, n_max) = 32 (hidden_dim 128 on graphs of 625-1231 nodes, round 4) with room in LDS for the
 * readout shares; GNM_ERR_UNSUPPORTED otherwise, before anything is launched (then gnm_bn_relu_readout followed by
 * gnm_agg). */
int gnm_agg_fwd_bnrelu(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off, const int64_t* b_col_off,
                       const int32_t* node_off, int B, int n_max, int nnz_max, const float* z, int ldz,
                       const float* scale, const float* shift, float* hout, int ldh, float* gf, int ldgf,
                       int graph_avg, float* y, int ldy, int F, const float* eps, int average, int self_loop,
                       void* stream);
int gnm_agg_slice_width(int F, int n_max);          /* feature-slice width the kernel will use (0: unsupported) */
int gnm_agg_num_partials(int F, int n_max, int B);  /* doubles written to deps_partial */
int gnm_sum_partials(const double* partial, int count, float* out, void* stream);

/* ---- neighbour aggregation on the matrix cores (dense graphs) --------------------------
 * The same three operations over a BIT adjacency: y = A x as MFMA products of the 0/1 matrix with three bf16 planes
 * of x (x split by truncation, every product exact: fp32-faithful like the gather).  Pays when the graphs are dense
 * (the 400-node benchmark graphs are 30 % dense: ~3x the gather); the caller chooses per batch.
 * Bit matrix of one graph with n nodes: W = ceil(n / 32) words = 4 W bytes of bits per row; bit (k % 8) of byte k / 8
 * = 1 iff k is a neighbour in row v of that graph's CSR.  A row's bytes are stored DE-INTERLEAVED (round 3): byte j in
 * half (j & 1) of the row at position (j >> 1) -- the even bytes are what lanes 0-31 of an MFMA step multiply, the odd
 * ones lanes 32-63, so a lane loads exactly its bytes -- each half padded with zeros to HP = ceil(W / 2) rounded up to 4
 * words; a row is 2 HP words and there are 32 W rows (zero rows pad the last block): gnm_adj_bits_words(n) = 64 W HP
 * words at adj_bits + b_bits_off[b] (adj_bits 16-byte aligned, offsets multiples of 4 words).  Built on the device from
 * the arena's CSR by gnm_adj_bits_build, which also
 * counts, per graph, CSR entries that repeat an edge (dup[g] > 0: a multigraph -- the bit matrix cannot carry the
 * multiplicity, keep that graph on gnm_agg).  Pass the bit matrix of the TRANSPOSED CSR for backward = 1.
 * All other arguments: exactly as in gnm_agg / gnm_agg_bwd_stats / gnm_agg_fwd_bnrelu (rowptr and the offsets are
 * still read: degrees).  deps_partial receives gnm_aggm_num_partials(F, B) doubles.
 * GNM_ERR_UNSUPPORTED (nothing launched; use the CSR form): n_max > gnm_aggm_max_nodes(); F neither a multiple of 32
 * with 16-byte aligned rows of x nor < 32 (F < 32, the input layer: gnm_aggm without d-eps only; the fused forms: F = 64). */
long long gnm_adj_bits_words(int n);
int gnm_aggm_max_nodes(void);
int gnm_aggm_num_partials(int F, int B);
int gnm_adj_bits_build(const int32_t* rowptr, const uint16_t* col, const int64_t* g_rp_off, const int64_t* g_col_off,
                       const int32_t* g_n, int G, uint32_t* bits, const int64_t* g_bits_off, int32_t* dup,
                       void* stream);
int gnm_aggm(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off, const int64_t* b_col_off,
             const uint32_t* adj_bits, const int64_t* b_bits_off, const int32_t* deg_rowptr, const int64_t* b_deg_off,
             const int32_t* node_off, int B, int n_max, const float* x, int ldx, float* y, int ldy, int F,
             const float* eps, int average, int self_loop, int backward, const float* hfwd, int ldh,
             double* deps_partial, void* stream);
int gnm_aggm_bwd_stats(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off, const int64_t* b_col_off,
                       const uint32_t* adj_bits, const int64_t* b_bits_off, const int32_t* deg_rowptr,
                       const int64_t* b_deg_off, const int32_t* node_off, int B, int n_max, const float* x, int ldx,
                       float* y, int ldy, int F, const float* eps, int average, int self_loop, const float* hfwd,
                       int ldh, double* deps_partial, const float* sZ, int ldsz, const float* s_scale,
                       const float* s_shift, const float* s_mean, const float* s_rstd, const float* dpool,
                       int ld_dpool, int graph_avg, const float* dsc1, const float* U, int ld_U,
                       const int32_t* inv_perm, const float* s2sum, double* s_partial, void* stream);
int gnm_aggm_fwd_bnrelu(const int32_t* rowptr, const uint16_t* col, const int64_t* b_rp_off, const int64_t* b_col_off,
                        const uint32_t* adj_bits, const int64_t* b_bits_off, const int32_t* node_off, int B, int n_max,
                        const float* z, int ldz, const float* scale, const float* shift, float* hout, int ldh,
                        float* gf, int ldgf, int graph_avg, float* y, int ldy, int F, const float* eps, int average,
                        int self_loop, void* stream);
/* d eps[l] = sum_v dpooled[v,:] . h[v,:] (graphcnn.py:161) without the gather, for a layer whose aggregation
 * backward has no other consumer: gnm_rowdot_num_partials() fp64 partials, to be summed like gnm_agg's. */
int gnm_rowdot_num_partials(void);
int gnm_rowdot_partials(const float* A, int lda, const float* B, int ldb, long long N, int F, double* partial,
                        void* stream);
/* nsets (<= 16) independent sets in one launch: out[k] = sum of partial[k*stride .. k*stride + counts_host[k]) */
int gnm_sum_partials_multi(const double* partial, long long stride, const int* counts_host, int nsets, float* out,
                           void* stream);

/* ---- Linear (mlp.py:25,32-35,43,48,49) on fp32 MFMA ---------------------------------
 * Z[N,H] = f(X)[N,K] W^T + bias with f(x) = x*pro_scale + pro_shift (then ReLU if
 * pro_relu) fused on load -- the BatchNorm+ReLU between two Linears (mlp.py:48).
 * w_kmajor = 0: W is torch's [H,K] weight;  1: W is [K,H] (used for dX = dZ W).
 * stats_partial (optional): [gnm_linear_grid(N)][2][H] doubles, per-column sum and sum
 * of squares of Z for the BatchNorm that follows.  H <= 128 per call. */
int gnm_linear_grid(int N);
/* Test hook (host arithmetic only): the first 32-row tile of wave `wave` of workgroup `b` in a launch of `nwg` workgroups
 * with `groups` groups of four waves each and `rows` groups in total -- the tile -> wave map of every tile-strided Linear
 * kernel (stride 4 x rows).  tests/test_host_logic.py checks that it is a bijection for every launch shape. */
int gnm_debug_lin_first_tile(int wave, int groups, int rows, int nwg, int b);
/* Largest input width K gnm_linear_fwd accepts for output width H (0: H unsupported; H <= 128).  The weight
 * stays LDS-resident, so K is bounded: 448 at H = 64, 192 at H = 128. */
int gnm_linear_max_k(int H);
int gnm_linear_fwd(const float* X, int ldx, const float* W, int ldw, int w_kmajor, const float* bias, float* Z,
                   int ldz, int N, int K, int H, const float* pro_scale, const float* pro_shift, int pro_relu,
                   double* stats_partial, void* stream);
/* dW[H,K] = dZ^T f(X), db[H] = column sums of dZ (autograd of nn.Linear). */
int gnm_wgrad_grid(int N);
long long gnm_wgrad_workspace_floats(int N, int H, int K);
int gnm_linear_wgrad(const float* dZ, int ldd, const float* X, int ldx, int N, int H, int K,
                     const float* pro_scale, const float* pro_shift, int pro_relu, float* dW, int ldw, float* db,
                     float* workspace, void* stream);

/* Fused backward of a Linear followed by BatchNorm (the autograd of mlp.py:48 / graphcnn.py:162-163):
 * dZ = cA*(G - m1 - xhat*m2) formed on the fly from G (output of gnm_bn_relu_bwd_stats) and Z,
 * then dX = dZ W (into dA, may be null), dW = dZ^T f(X), db = sum dZ -- one pass instead of
 * gnm_bn_bwd_apply + gnm_linear_wgrad + gnm_linear_fwd(w_kmajor=1).  Eligible for K, H in {32, 64} and
 * 16-B aligned rows; otherwise returns GNM_ERR_UNSUPPORTED without launching anything.
 * workspace: gnm_linear_bwd_workspace_floats(N, H, K) floats. */
int gnm_linear_bwd_grid(int N);
long long gnm_linear_bwd_workspace_floats(int N, int H, int K);
int gnm_linear_bwd_fused(const float* G, int ldg, const float* Z, int ldz, const float* mean, const float* rstd,
                         const float* cA, const float* m1, const float* m2, const float* X, int ldx,
                         const float* pro_scale, const float* pro_shift, int pro_relu, const float* W, int ldw,
                         float* dA, int lda, float* dW, int lddw, float* db, float* workspace, int N, int K, int H,
                         const float* sZ, int ldsz, const float* s_scale, const float* s_shift, const float* s_mean,
                         const float* s_rstd, double* s_partial, void* stream);
/* The same pass for a Linear whose stored output the caller does not hand over: Z = f(X) W^T + bias (what
 * gnm_linear_fwd wrote, mlp.py:43,49) is recomputed in the kernel instead of being read -- three [N,64] streams instead
 * of four.  K = H = 64 and dA wanted; sZ either NULL or == X with (s_scale, s_shift) == (pro_scale, pro_shift) and
 * pro_relu (the two Linears of a 2-layer MLP).  Anything else: GNM_ERR_UNSUPPORTED, nothing launched -- call
 * gnm_linear_bwd_fused with Z.  Workspace, partial rows and the dW = NULL deferral as for gnm_linear_bwd_fused. */
int gnm_linear_bwd_fused_rz(const float* G, int ldg, const float* bias, const float* mean, const float* rstd,
                            const float* cA, const float* m1, const float* m2, const float* X, int ldx,
                            const float* pro_scale, const float* pro_shift, int pro_relu, const float* W, int ldw,
                            float* dA, int lda, float* dW, int lddw, float* db, float* workspace, int N, int K, int H,
                            const float* sZ, int ldsz, const float* s_scale, const float* s_shift, const float* s_mean,
                            const float* s_rstd, double* s_partial, void* stream);

/* dX = dZ W of a K = H = 128 Linear whose input came through BatchNorm + ReLU (mlp.py:48), with that ReLU's mask and that
 * BatchNorm's backward sums in the epilogue (replaces gnm_linear_fwd(w_kmajor = 1) + gnm_bn_relu_bwd_stats for the inner
 * BatchNorms of an H = 128 model):  G[n,k] = (dZ W)[n,k] where mZ[n,k] m_scale[k] + m_shift[k] > 0, else 0;
 * stats_partial [gnm_linear_grid(N)][2][K]: partial sums of G and of G (mZ - m_mean) m_rstd, as gnm_bn_bwd_finalize takes
 * them.  W [H][K] row-major (the Linear's weight).  GNM_ERR_UNSUPPORTED for other shapes. */
int gnm_linear_dgrad_masked(const float* dZ, int ldd, const float* W, int ldw, float* G, int ldg, int N, int K, int H,
                            const float* mZ, int ldmz, const float* m_scale, const float* m_shift, const float* m_mean,
                            const float* m_rstd, double* stats_partial, void* stream);
/* dW = NULL defers the reduction of the per-workgroup dW / db partials: they stay in `workspace` (keep it alive and
 * unshared) until ONE gnm_reduce_partials_multi call reduces up to 32 such workspaces of the same N, e.g. at the end
 * of a backward pass (nothing in the backward reads a weight gradient).  HOST arrays of njobs entries. */
int gnm_reduce_partials_multi(const float* const* workspaces_host, float* const* dW_host, const int* lddw_host,
                              float* const* db_host, const int* Hs_host, const int* Ks_host, int njobs, int N,
                              void* stream);

/* The [B, L*H]-sized matrix products of the Infomax tail (discriminator.py:30-31 through nn.Bilinear, restructured as
 * U = sigmoid(g_f) Wd^T with backward dWd = dU^T sigmoid(g_f), T = dU Wd): C[M,N] = A' B', fp32 in and out, fp32-accurate
 * (split-precision bf16 products), fixed summation order.  A' = A ([M][K], a_cols = 0) or A^T (A given as [K][M],
 * a_cols = 1); B' = B^T (B given as [N][K], b_cols = 0) or B ([K][N], b_cols = 1).  GNM_ERR_UNSUPPORTED (nothing
 * launched) when an operand exceeds a 32-bit byte offset. */
int gnm_small_gemm(const float* A, int lda, int a_cols, const float* B, int ldb, int b_cols, float* C, int ldc, int M, int N,
                   int K, void* stream);
/* sZ (optional): dA is the gradient arriving at relu(bn_lo(sZ)), the BatchNorm+ReLU feeding this Linear
 * (mlp.py:48).  Then dA is written already multiplied by that ReLU mask and s_partial receives
 * [gnm_linear_bwd_grid(N)][2][K] doubles (sum g, sum g*xhat) for gnm_bn_bwd_finalize -- i.e. the call also
 * replaces gnm_bn_relu_bwd_stats for the lower BatchNorm. */

/* ---- BatchNorm1d + ReLU + readout (mlp.py:38,48; graphcnn.py:51,163-166,187-190,228-229) */
int gnm_bn_finalize(const double* stats_partial, int nblk, int H, long long nrows, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                    float momentum, float eps, int training, int update_running, float* scale, float* shift,
                    float* mean_out, float* rstd_out, void* stream);
/* Hout = relu(Z*scale+shift) (may be null: only the readout is wanted); pooled[b] = sum (or mean) of graph b's
 * rows (may be null). */
int gnm_bn_relu_readout(const float* Z, int ldz, const float* scale, const float* shift, float* Hout, int ldh,
                        const int32_t* node_off, int B, int H, int relu, float* pooled, int ldp, int average,
                        void* stream);
/* X_concat (/root/reference models/graphcnn.py:195): dst[node_off[b] + r, :width] = src[base[b] + r, :width] for the n_b
 * rows of every graph b of the batch (base: int64 [B], first row of the graph where the arena keeps its features);
 * src2 / dst2 (may be null): a second array of the same shape copied alongside. */
int gnm_gather_graph_rows(const float* src, const float* src2, int lds, int width, const long long* base,
                          const int32_t* node_off, int B, float* dst, float* dst2, int ldd, void* stream);
/* Backward pass 1: G = (dH + readout grad + discriminator grads) * relu mask, and the
 * per-graph (sum G, sum G*xhat) partials [B][2][H]. */
int gnm_bn_relu_bwd_stats(const float* dH, int lddh, const float* dpool, int ldp, int average, const float* dsc1,
                          const float* U, int ldu, const int32_t* inv_perm, const float* s2sum, const float* Z,
                          int ldz, const float* scale, const float* shift, const float* mean, const float* rstd,
                          int relu, float* G, int ldg, const int32_t* node_off, int B, int H, double* partial,
                          void* stream);
int gnm_bn_bwd_finalize(const double* partial, int nblk, int H, long long nrows, const float* gamma,
                        const float* rstd, int training, float* dgamma, float* dbeta, float* cA, float* m1,
                        float* m2, void* stream);
/* Backward pass 2: dZ = cA * (G - m1 - xhat*m2); dZ may alias G. */
int gnm_bn_bwd_apply(const float* G, int ldg, const float* Z, int ldz, const float* mean, const float* rstd,
                     const float* cA, const float* m1, const float* m2, float* dZ, int ldd, long long N, int H,
                     void* stream);

/* ---- one-launch evaluation encoder (graphcnn.py:208-231 in eval() mode; main.py:49-57, 71-82) ----------
 * The L GIN layers (aggregation -> MLP -> BatchNorm on its RUNNING statistics -> ReLU), the per-layer graph readout and
 * the classifier head of B graphs in ONE launch, one workgroup per graph (csrc/evalfwd.hip): what the reference's
 * per-graph evaluation loop calls once per graph.
 * table: DEVICE array of gnm_eval_table_words(L, m) int64 words with the parameter addresses -- entry l * m + k (7 words:
 * W, bias, gamma, beta, running_mean, running_var, then the weight's leading dimension) for Linear k of layer l's MLP
 * and the BatchNorm BEHIND it (the MLP's inner BatchNorm k for k < m - 1, the layer's outer BatchNorm for k = m - 1),
 * followed by 2 words per layer (classifier weight [C, H] row-major, classifier bias [C]).
 * eps: [L] or NULL (learn_eps False: self loops).  hidden: OUTPUT, the L hidden layers [L][N, H] (layer stride
 * hidden_stride floats, leading dimension ldh); s0 / s1: two [N, lds] fp32 scratch arrays.  Outputs: hidden, g_f
 * [B, L * H], c_sig = sigmoid(g_f) (may be NULL), c_logit [B, C].
 * GNM_ERR_UNSUPPORTED (nothing launched; run the layer-by-layer entry points): H != 64, m > 3, L > 16, F0 > 64, C > 64,
 * a graph of more than gnm_eval_max_nodes() nodes (every graph needs a bit adjacency).  Arithmetic: the training kernels'
 * (three-plane bf16 splits, fp32 accumulation); results agree with them to fp32 rounding. */
int gnm_eval_max_nodes(void);
long long gnm_eval_table_words(int L, int m);
int gnm_eval_encoder(const uint32_t* adj_bits, const int64_t* b_bits_off, const int32_t* node_off, const int32_t* rowptr,
                     const int64_t* b_rp_off, int B, int n_max, const float* X, int ldx, int F0, int H, int L, int m, int C,
                     int average, int self_loop, int graph_avg, float bn_eps, const long long* table, const float* eps,
                     float* hidden, long long hidden_stride, int ldh, float* s0, float* s1, int lds, float* g_f, int ldgf,
                     float* c_sig, float* c_logit, int ldc, void* stream);

/* The same eval-mode encoder + readout + classifier as L + 1 launches: one launch per GIN layer with a workgroup per
 * 32-row block of every graph (13 CUs work on one 400-node graph), then the readout sums + classifier head
 * (graphcnn.py:208-231, main.py:49-57).  Arguments as gnm_eval_encoder, with `scratch`
 * (gnm_eval_layers_scratch_floats(B, n_max, H, L) floats) in place of its two [N, H] arrays.  H in {32, 64, 128},
 * 1 <= m <= 3, F0 <= 128, C <= 256, n_max <= 416, every graph with a bit adjacency: GNM_ERR_UNSUPPORTED otherwise. */
long long gnm_eval_layers_scratch_floats(int B, int n_max, int H, int L);
int gnm_eval_layers(const uint32_t* adj_bits, const int64_t* b_bits_off, const int32_t* node_off, const int32_t* rowptr,
                    const int64_t* b_rp_off, int B, int n_max, const float* X, int ldx, int F0, int H, int L, int m, int C,
                    int average, int self_loop, int graph_avg, float bn_eps, const long long* table, const float* eps,
                    float* hidden, long long hidden_stride, int ldh, float* scratch, float* g_f, int ldgf, float* c_sig,
                    float* c_logit, int ldc, void* stream);

/* ---- Infomax discriminator (discriminator.py:19-38, graphcnn.py:233-246) ------------
 * hptrs_host: HOST array of L device pointers to the per-layer [N,H] hidden states
 * (n_f is never concatenated).  A layer may instead be given as the pre-BatchNorm output Z_l of its last Linear
 * plus the folded BatchNorm vectors: scale_ptrs_host[l] / shift_ptrs_host[l] non-NULL (HOST arrays of L device
 * pointers to [H] vectors, or NULL arrays) make the kernels use relu(hptrs[l] * scale + shift) (graphcnn.py:163-166),
 * so the activation never has to exist in memory (gnm_agg_fwd_bnrelu with hout = NULL).
 * U = sigmoid(g_f) W^T, [B, L*H].  perm_rows[g] = perm[g], the ROW of n_f the reference's shuffle index selects
 * for graph g.  d_logit: [2N] (= the reference's [2N,1]). */
int gnm_disc_score_fwd(const float* const* hptrs_host, const float* const* scale_ptrs_host,
                       const float* const* shift_ptrs_host, int ldh, int L, int H, const float* U, int ldu,
                       const int32_t* perm_rows, const float* bias, const int32_t* node_off, int N, int B,
                       float* d_logit, void* stream);
/* gnm_disc_score_fwd that ALSO leaves the reductions its backward needs, for the loss the reference applies to
 * d_logit -- BCEWithLogitsLoss against ones for the first N entries and zeros for the rest (main.py:32-37) -- up to that
 * loss's scalar factor k (= upstream gradient * beta / 2N):
 *   unit[g, l*H + c] = dU[g, l*H + c] / k,  unit[g, L*H] = s2sum[g] / k,  unit[g, L*H + 1] = dsum[g] / k   (ldunit >=
 *   L*H + 2, a multiple of 4, 16-byte aligned base),  inv_perm[perm_rows[g]] = g.
 * They come from the hidden rows the score kernel holds in registers anyway, so gnm_disc_score_bwd's second pass over
 * the L hidden layers (539 MB at B = 1024) is not needed: the backward calls gnm_disc_unit_scale with k.  A caller whose
 * loss on d_logit is anything else ignores `unit` and calls gnm_disc_score_bwd as before.
 * GNM_ERR_UNSUPPORTED outside the vector forms (H / 4 in {8, 16, 32}, L <= 5, ldh % 4 == 0). */
int gnm_disc_score_fwd_unit(const float* const* hptrs_host, const float* const* scale_ptrs_host,
                            const float* const* shift_ptrs_host, int ldh, int L, int H, const float* U, int ldu,
                            const int32_t* perm_rows, const float* bias, const int32_t* node_off, int N, int B,
                            float* d_logit, float* unit, int ldunit, int32_t* inv_perm, void* stream);
/* dU = k * unit[:, :LH], s2sum = k * unit[:, LH], dsum = k * unit[:, LH + 1] (dsum may be NULL); k: DEVICE scalar
 * (times the host factor kscale).  dbias (may be NULL): one float, the sum of dsum over the B graphs = the gradient of
 * the Bilinear's bias (/root/reference models/discriminator.py:19), added in a fixed order by the same launch. */
int gnm_disc_unit_scale(const float* unit, int ldunit, int LH, const float* k, float kscale, int B, float* dU, int ldu,
                        float* s2sum, float* dsum, float* dbias, void* stream);
/* optional by-products: dsum[g] = sum over graph g of dD (both halves; their total is d bias), and
 * inv_perm[perm_rows[g]] = g. */
int gnm_disc_score_bwd(const float* const* hptrs_host, const float* const* scale_ptrs_host,
                       const float* const* shift_ptrs_host, int ldh, int L, int H, const float* dD,
                       const int32_t* perm_rows, const int32_t* node_off, int N, int B, float* dU, int ldu,
                       float* s2sum, float* dsum, int32_t* inv_perm, void* stream);

/* ---- graph-level head (graphcnn.py:224-231, 239) --------------------------------------
 * wp_host / bp_host: HOST arrays of L device pointers to linears_prediction[l].weight ([C,H] row-major) and
 * .bias.  masks: the dropout masks [L,B,C] (0 or 1/(1-p)) or NULL.
 * gnm_head_fwd:  c_logit[b,c] = sum_l masks[l,b,c] * (g_f[b, lH:(l+1)H] . W_l[c,:] + b_l[c])   (:230, score_over_layer)
 *                csig = sigmoid(g_f)                                                           (:239, may be NULL)
 * gnm_head_bwd:  dph [B, L*H] = d loss / d g_f = classifier path + T * csig * (1 - csig) (T = dU Wd or NULL);
 *                dwp_host / dbp_host: HOST arrays of L device pointers receiving the classifier gradients.
 * GNM_ERR_UNSUPPORTED when C > 256 or L > 16 (the caller then uses plain matrix products). */
int gnm_head_fwd(const float* g_f, int ldg, int B, int L, int H, int C, const float* const* wp_host,
                 const float* const* bp_host, const float* masks, float* c_logit, int ldc, float* csig, int ldcs,
                 void* stream);
int gnm_head_bwd(const float* dC, int lddc, const float* masks, const float* g_f, int ldg, const float* csig,
                 int ldcs, const float* T, int ldt, int B, int L, int H, int C, const float* const* wp_host,
                 float* const* dwp_host, float* const* dbp_host, float* dph, int lddph, void* stream);

/* ---- neighbor_pooling_type == "max" (csrc/maxpool.hip) ---------------------------------
 * Replaces __preprocess_neighbors_maxpool + maxpool (graphcnn.py:55-81, 137-143) and the autograd of
 * torch.max(h_with_dummy[padded_neighbor_list], dim = 1) behind them.  The padded [N, max_deg (+1)] list is not
 * built: nb_off [N+1] / nb_col are the concatenated graph.neighbors lists (batch-global row ids, the lists' own
 * order), max_deg the batch maximum (graph.max_neighbor, :59); a row with fewer neighbours gets ONE dummy candidate
 * (`dummy` [F] = column minimum of h, :140; may be NULL when no row is shorter than max_deg) after them, and self_last = 1 (learn_eps False, :73-74) appends the row
 * itself.  Selection follows ATen's CPU scan (first maximum wins, a NaN ends the scan); amax [N*F] (optional)
 * records the selected row per element (-1 = the dummy), which is where the backward sends the gradient.
 * eps != NULL: out = max + (1 + *eps) * h (graphcnn.py:161).  Returns GNM_ERR_BAD_ARG where torch raises (no
 * candidate at all: max_deg == 0 without self_last).
 * gnm_maxpool_colmin: vmin [F] / amin [F] = torch.min(h, dim = 0) values and (first-occurrence) rows; ws_val /
 * ws_idx hold gnm_maxpool_colmin_blocks(N) * F entries each.
 * gnm_maxpool_bwd: dh[j] = sum of g[i] over the rows i whose amax is j (t_off [N+1] / t_col: for every row j the
 * DISTINCT rows i that have j as a candidate, ascending -- include (j, j) when self_last was set) + (1 + *eps) g[j];
 * iso_rows [n_iso] = rows without neighbours (the only ones that can select the dummy): their gradient goes to row
 * amin[c].  No atomics: results are bitwise repeatable.
 * The _tiled forms do the same work with one workgroup per graph and the graph's rows staged in LDS (node_off [B+1]
 * = first row of each graph, n_max = most rows of one graph): same candidates, same scan order, bitwise the same
 * results, ~15x faster on 400-node graphs.  They take F = 32 or 64 with 16-byte aligned rows and a tile that fits the
 * CU's LDS (about 4 F n_max bytes forward, 6 F n_max backward) and return GNM_ERR_UNSUPPORTED otherwise. */
int gnm_maxpool_colmin_blocks(int N);
int gnm_maxpool_colmin(const float* h, int ldh, int N, int F, float* ws_val, int32_t* ws_idx, float* vmin, int32_t* amin,
                       void* stream);
int gnm_maxpool_fwd(const float* h, int ldh, const int32_t* nb_off, const int32_t* nb_col, int N, int F, int max_deg,
                    int self_last, const float* eps, const float* dummy, float* out, int ldo, int32_t* amax, void* stream);
int gnm_maxpool_bwd(const float* g, int ldg, const int32_t* amax, const int32_t* t_off, const int32_t* t_col, int N, int F,
                    const float* eps, const int32_t* iso_rows, int n_iso, const int32_t* amin, float* dh, int ldd,
                    void* stream);
int gnm_maxpool_fwd_tiled(const float* h, int ldh, const int32_t* nb_off, const int32_t* nb_col, const int32_t* node_off,
                          int B, int n_max, int F, int max_deg, int self_last, const float* eps, const float* dummy,
                          float* out, int ldo, int32_t* amax, void* stream);
int gnm_maxpool_bwd_tiled(const float* g, int ldg, const int32_t* amax, const int32_t* t_off, const int32_t* t_col,
                          const int32_t* node_off, int B, int n_max, int F, const float* eps, const int32_t* iso_rows,
                          int n_iso, const int32_t* amin, float* dh, int ldd, void* stream);

/* ---- train-step tail ----------------------------------------------
 * gnm_loss_ce_bce replaces, in the reference's train() (main.py:16-17, 32-37):
 *     c_loss = CrossEntropyLoss()(c_logit, c_labels)
 *     d_loss = BCEWithLogitsLoss()(d_logit, d_labels)
 *     loss   = c_loss + beta * d_loss                 and the first step of loss.backward()
 * loss3 = {loss, c_loss, d_loss}; dC [B,C] = d loss / d c_logit; dD [M] = d loss / d d_logit (either may be
 * NULL).  labels: int64 [B], each in [0,C).  d_target [M] or NULL = the reference's labels (first n_pos
 * entries 1, the rest 0; main.py:32).  workspace: gnm_loss_workspace_doubles(M) doubles. */
long long gnm_loss_workspace_doubles(long long M);
int gnm_loss_ce_bce(const float* c_logit, int ldc, const long long* labels, int B, int C, const float* d_logit,
                    const float* d_target, long long M, long long n_pos, float beta, float* loss3, float* dC,
                    int lddc, float* dD, double* workspace, void* stream);
/* The same gradients without the loss values, multiplied by the upstream gradient *gscale_dev (device scalar;
 * NULL = 1): what loss.backward() needs, in one launch. */
int gnm_loss_ce_bce_grad(const float* c_logit, int ldc, const long long* labels, int B, int C, const float* d_logit,
                         const float* d_target, long long M, long long n_pos, float beta, const float* gscale_dev,
                         float* dC, int lddc, float* dD, void* stream);
/* gnm_adam_step replaces optimizer.step() of optim.Adam(model.parameters(), lr) (main.py:136, 39-41) on a flat
 * fp32 parameter buffer: torch.optim.Adam's default update (no AMSGrad, L2 weight decay).
 * hyper: DEVICE array of 6 doubles {lr, beta1, beta2, eps, weight_decay, grad_scale} (grad is multiplied by
 * grad_scale first: 1/world after a sum all-reduce); step: DEVICE int32, updates done so far, incremented on
 * the stream after the update -- so the call is hipGraph-capturable and StepLR (main.py:137,153) is a write
 * of hyper[0] between replays. */
int gnm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                  const double* hyper, int32_t* step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GNM_HIP_H */
