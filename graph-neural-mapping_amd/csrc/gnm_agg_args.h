// Argument block shared by the aggregation kernels: the CSR gather kernels (agg.hip) and the matrix-core kernel over
// the bit adjacency (aggm.hip).  One struct, so the three C-ABI entry points fill it once and either path takes it.
#pragma once
#include "gnm_common.h"

struct AggArgs {
    const int32_t* rowptr;     // gather structure arena (forward CSR, or transposed for backward)
    const uint16_t* col;       // graph-local column ids
    const int64_t* b_rp_off;   // [B] offset of graph b's rowptr block
    const int64_t* b_col_off;  // [B] offset of graph b's col block
    const int32_t* deg_rowptr; // forward CSR rowptr arena (degrees for the backward pre-scale)
    const int64_t* b_deg_off;  // [B]
    const int32_t* node_off;   // [B+1] first row of each graph in the batch
    const float* x;
    float* y;
    const float* eps;          // device pointer to eps[layer], or null
    const float* hfwd;         // backward: forward input of the layer (for d eps), or null
    double* deps_partial;      // [B * nslices] or null
    int ldx, ldy, ldh;
    int F;                     // valid feature width
    int nslices;
    int average, self_loop, backward;
    // optional fusion (agg16, backward, one slice): y is the gradient arriving at relu(bn(sZ)) of the layer
    // below -- add the readout / discriminator terms, apply that ReLU mask, write G and reduce the
    // BatchNorm-backward sums (replaces gnm_bn_relu_bwd_stats for that BatchNorm)
    const float* sZ; const float* s_scale; const float* s_shift; const float* s_mean; const float* s_rstd;
    const float* s_dpool; const float* s_dsc1; const float* s_U; const int32_t* s_inv_perm; const float* s_s2sum;
    double* s_partial;         // [B][2][64]
    int ldsz, ld_dpool, ld_U, s_avg, n_batch;
    int ids_in_lds;            // narrow slices: the graph's column ids are staged in LDS (max_nnz given)
    int debug;                 // tuning only (GNM_AGG16_DEBUG): 1 no id loads, 2 no epilogue/store, 4 no combine
    unsigned long long* stamps;   // tuning only: per-wave s_memtime stamps, [workgroup][16 waves][64] (gnm_debug_set_stamps)
    // forward prologue (agg16 only): the input is Z of the previous layer's last Linear; the tile load applies that
    // layer's outer BatchNorm + ReLU, writes the activation h (p_hout) and its graph readout (p_gf) on the way
    const float* p_scale; const float* p_shift;
    float* p_hout; float* p_gf;
    int p_ldh, p_ldgf, p_gf_avg;
    // bit adjacency (aggm.hip only): graph b's [32 W][W] words, W = ceil(n / 32), at adj_bits + b_bits_off[b]
    const uint32_t* adj_bits; const int64_t* b_bits_off;
    int n_graphs, n16_max;     // aggm.hip: B (the grid is padded to whole XCD rounds) and ceil(n_max / 16) * 16
};
