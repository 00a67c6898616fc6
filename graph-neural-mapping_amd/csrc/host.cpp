// Host-side helpers of the C-ABI: per-graph CSR construction from the reference's
// edge_mat ([2, E] int64, util.py:99-103) -- the device-resident replacement for
// GIN_InfoMaxReg.__preprocess_neighbors_sumavepool (/root/reference
// models/graphcnn.py:84-106) -- and the inverse expansion used by the parity tests to
// prove the CSR encodes the identical edge multiset and offsets.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gnm_once.h"

extern "C" {

const char* gnm_version(void) { return "gnm_hip 0.2 (gfx950)"; }

// Test hook for the per-device launch-configuration guard (GnmDeviceOnce, the object behind
// GNM_ALLOW_FULL_LDS): behaves exactly like one kernel's guard, on a private instance and without
// touching any device.  Returns 1 when `device` had not been configured yet (and marks it), else 0;
// reset != 0 forgets every device first.
int gnm_debug_device_once(int device, int reset) {
    static GnmDeviceOnce once;
    if (reset)
        for (int d = 0; d < kGnmMaxDevices; ++d) once.done[d].store(0);
    if (!once.first_use(device)) return 0;
    once.mark(device);
    return 1;
}

// rowptr[n+1], col[E]: row = edge_mat[0][e], col = edge_mat[1][e]; edges of a row keep
// their edge_mat order (stable counting sort); duplicates are kept (COO duplicates add).
int gnm_csr_from_edge_mat(const int64_t* edge_mat, long long E, int n, int32_t* rowptr, uint16_t* col) {
    if (n < 0 || n > 65535 || E < 0) return -1;
    const int64_t* src = edge_mat;
    const int64_t* dst = edge_mat + E;
    for (int i = 0; i <= n; ++i) rowptr[i] = 0;
    for (long long e = 0; e < E; ++e) {
        if (src[e] < 0 || src[e] >= n || dst[e] < 0 || dst[e] >= n) return -2;
        rowptr[src[e] + 1]++;
    }
    for (int i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    std::vector<int32_t> fill(rowptr, rowptr + n);
    for (long long e = 0; e < E; ++e) col[fill[src[e]]++] = (uint16_t)dst[e];
    return 0;
}

int gnm_csr_transpose(const int32_t* rowptr, const uint16_t* col, int n, int32_t* rowptr_t, uint16_t* col_t) {
    const int32_t E = rowptr[n];
    for (int i = 0; i <= n; ++i) rowptr_t[i] = 0;
    for (int32_t e = 0; e < E; ++e) rowptr_t[col[e] + 1]++;
    for (int i = 0; i < n; ++i) rowptr_t[i + 1] += rowptr_t[i];
    std::vector<int32_t> fill(rowptr_t, rowptr_t + n);
    for (int r = 0; r < n; ++r)
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) col_t[fill[col[e]]++] = (uint16_t)r;
    return 0;
}

// Order every CSR row's ids for the 32-float-slice gather (csrc/agg.hip, gnm_agg_kernel<8>): position j of a row wants
// an EVEN id when (j & 3) < 2 and an ODD one otherwise, as long as the row has both kinds left.  Why: that kernel reads
// eight neighbour rows of 128 bytes per ds_read_b128; the hardware serves the instruction in 16-lane groups that pair
// the neighbours at positions (8s, 8s+3), (8s+1, 8s+2), (8s+4, 8s+7), (8s+5, 8s+6), and two 128-byte LDS rows of EQUAL
// parity lie on the same 32 banks (round 3 PMC: a third of the kernel's LDS cycles were such conflicts).  The edge
// multiset of a row is unchanged (a sum does not care; graphcnn.py:91-104 fixes no order either) and the relative order
// inside each parity class is kept.  Every other kernel is indifferent to the order.
int gnm_csr_parity_order(const int32_t* rowptr, uint16_t* col, int n) {
    if (n < 0) return -1;
    std::vector<uint16_t> ev, od;
    for (int r = 0; r < n; ++r) {
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        if (e - b < 2) continue;
        ev.clear(); od.clear();
        for (int32_t k = b; k < e; ++k) (col[k] & 1 ? od : ev).push_back(col[k]);
        size_t ie = 0, io = 0;
        for (int32_t k = b; k < e; ++k) {
            const bool want_odd = ((k - b) & 3) >= 2;
            const bool take_odd = want_odd ? io < od.size() : ie >= ev.size();
            col[k] = take_odd ? od[io++] : ev[ie++];
        }
    }
    return 0;
}

// 1 when A and A^T hold the same edge multiset (then the backward may gather over A itself).
// Two counting-sort transposes, O(n + E): T1 = A^T and T2 = T1^T = A both come out with sorted rows, so A is
// symmetric exactly when the two canonical forms are equal (round 1 sorted every row: 1/3 of the 9 ms first touch).
int gnm_csr_is_symmetric(const int32_t* rowptr, const uint16_t* col, int n) {
    const int32_t E = rowptr[n];
    std::vector<int32_t> rp1(n + 1), rp2(n + 1);
    std::vector<uint16_t> c1(E > 0 ? E : 1), c2(E > 0 ? E : 1);
    gnm_csr_transpose(rowptr, col, n, rp1.data(), c1.data());
    if (memcmp(rp1.data(), rowptr, sizeof(int32_t) * (n + 1)) != 0) return 0;     // in-degrees != out-degrees
    gnm_csr_transpose(rp1.data(), c1.data(), n, rp2.data(), c2.data());
    return memcmp(c1.data(), c2.data(), sizeof(uint16_t) * (size_t)E) == 0 ? 1 : 0;
}

// Expand a batch of per-graph CSRs to the reference's block-diagonal COO index
// (Adj_block._indices(), graphcnn.py:91-104): out[0][e] = row + start[b],
// out[1][e] = col + start[b], row-major by (graph, row, CSR position); when
// self_loops != 0 the N self loops are appended as the reference does (:97-102).
// Returns the number of entries written.
long long gnm_batch_coo_from_csr(const int32_t* rowptr_arena, const uint16_t* col_arena, const int64_t* b_rp_off,
                                 const int64_t* b_col_off, const int32_t* node_off, int B, int self_loops,
                                 int64_t* out_rows, int64_t* out_cols) {
    long long k = 0;
    for (int b = 0; b < B; ++b) {
        const int32_t* rp = rowptr_arena + b_rp_off[b];
        const uint16_t* cl = col_arena + b_col_off[b];
        const int n = node_off[b + 1] - node_off[b];
        for (int r = 0; r < n; ++r)
            for (int32_t e = rp[r]; e < rp[r + 1]; ++e) {
                out_rows[k] = (int64_t)node_off[b] + r;
                out_cols[k] = (int64_t)node_off[b] + cl[e];
                ++k;
            }
    }
    if (self_loops) {
        const int N = node_off[B];
        for (int v = 0; v < N; ++v) {
            out_rows[k] = v;
            out_cols[k] = v;
            ++k;
        }
    }
    return k;
}

}  // extern "C"
