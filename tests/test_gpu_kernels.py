"""Kernel-level GPU parity through the C-ABI (libgnm_hip.so), against the CPU oracle's op
restatements (oracle/gin_oracle.py) evaluated in float64 on the same seeded inputs, plus
size-independent properties at BASELINE.json's full sizes (n = 400, E = 47,600, B = 1024).

Tolerance: 1e-5 relative (max-norm) for every fp32 kernel, written at each check."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import assert_close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def _stream():
    return torch.cuda.current_stream().cuda_stream


class RG:
    """random graph object with the S2VGraph fields the arena reads"""

    def __init__(self, n, edge_mat, f0, rng):
        self.g = range(n)
        self.edge_mat = torch.from_numpy(np.ascontiguousarray(edge_mat.astype(np.int64)))
        self.node_features = torch.from_numpy(rng.standard_normal((n, f0)).astype(np.float32))
        self.label = 0


def random_graphs(rng, sizes, density, symmetric=True, f0=3):
    out = []
    for n in sizes:
        A = rng.random((n, n)) < density
        np.fill_diagonal(A, False)
        if symmetric:
            A = np.triu(A, 1)
            A = A | A.T
        src, dst = np.nonzero(A)
        out.append(RG(n, np.stack([src, dst]), f0, rng))
    return out


def dense_adj(batch_graphs):
    """block-diagonal dense adjacency (float64), the oracle's coo_to_csr semantics"""
    from oracle import gin_oracle as O
    ob = [O.OGraph(len(g.g), g.edge_mat.numpy(), g.node_features.numpy()) for g in batch_graphs]
    idx, val, shape = O.build_adj_block(ob, learn_eps=True)
    return O.coo_to_csr(idx, val, shape, np.float64)


AGG_CASES = [
    # (sizes, density, F, symmetric)
    ([40, 40, 40], 0.3, 64, True),        # headline shape class: LPR 16
    ([37, 5, 64, 1, 23], 0.4, 64, True),  # ragged sizes incl. a single-node graph
    ([50, 50], 0.0, 64, True),            # no edges at all
    ([200, 200], 1.0, 64, True),          # complete graphs: degree 199 -> third id chunk fetched in place
    ([30, 31], 0.3, 7, True),             # layer-0 width: LPR 2, scalar loads
    ([30, 31], 0.3, 5, True),
    ([64, 64], 0.2, 32, True),            # LPR 8
    ([48, 48], 0.3, 16, True),            # LPR 4
    ([300, 300], 0.05, 128, True),        # LPR 32
    ([1000, 1000], 0.02, 128, True),      # config-4 class: tile does not fit -> 4 feature slices of 32
    ([400, 400], 0.3, 400, True),         # one-hot width of the reference's default input: 7 slices of 64
    ([45, 45, 45], 0.3, 64, False),       # asymmetric
    ([45, 45, 45], 0.3, 20, False),       # asymmetric, partial slice, unaligned rows
]


@pytest.mark.parametrize("sizes,density,F,symmetric", AGG_CASES)
@pytest.mark.parametrize("average,learn_eps", [(0, 1), (1, 1), (0, 0), (1, 0)])
def test_agg_forward_backward(sizes, density, F, symmetric, average, learn_eps):
    from gnm import core
    from gnm.arena import GraphArena
    rng = np.random.default_rng(hash((tuple(sizes), F)) % 2**31)
    graphs = random_graphs(rng, sizes, density, symmetric)
    ar = GraphArena(DEV)
    batch = ar.batch(graphs)
    batch.dense = False              # this test is about the CSR gather kernels (the matrix-core path: test_gpu_aggm.py)
    assert batch.symmetric == (symmetric or density == 0.0)
    A = dense_adj(graphs)
    N = batch.N
    deg = np.asarray(A.sum(1)).reshape(-1, 1)
    x = rng.standard_normal((N, F))
    eps = 0.37
    spec = core.GinSpec(1, 1, bool(learn_eps), "sum", "average" if average else "sum")
    xd = torch.from_numpy(x.astype(np.float32)).to(DEV)
    yd = torch.full((N, F), float("nan"), device=DEV)
    epsd = torch.tensor([eps], device=DEV)
    core._agg(batch, xd, yd, F, epsd.data_ptr() if learn_eps else None, spec, backward=False)
    x32 = xd.cpu().numpy().astype(np.float64)
    # forward reference (graphcnn.py:154-161 / 178-182)
    with np.errstate(divide="ignore", invalid="ignore"):
        if learn_eps:
            ref = A @ x32
            if average:
                ref = ref / deg
            ref = ref + (1 + np.float32(eps).astype(np.float64)) * x32
        else:
            ref = A @ x32 + x32
            if average:
                ref = ref / (deg + 1)
    assert_close(yd.cpu().numpy(), ref, rtol=TOL, what="agg forward")
    # backward: dh = A^T (dp * pre) + self terms; deps = sum dp * h
    dp = rng.standard_normal((N, F)).astype(np.float32)
    dpd = torch.from_numpy(dp).to(DEV)
    dhd = torch.full((N, F), float("nan"), device=DEV)
    part = torch.empty(core.lib.gnm_agg_num_partials(F, batch.n_max, batch.B), dtype=torch.float64, device=DEV)
    core._agg(batch, dpd, dhd, F, epsd.data_ptr() if learn_eps else None, spec, backward=True,
              hfwd=xd if learn_eps else None, deps_partial=part if learn_eps else None)
    dp64 = dp.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        if learn_eps:
            src = dp64 / deg if average else dp64
            src = np.where(np.isfinite(src), src, 0.0)      # isolated rows are never gathered
            refb = A.T @ src + (1 + np.float32(eps).astype(np.float64)) * dp64
        else:
            src = dp64 / (deg + 1) if average else dp64
            refb = A.T @ src + src
    assert_close(dhd.cpu().numpy(), refb, rtol=TOL, what="agg backward")
    if learn_eps:
        out = torch.empty(1, device=DEV)
        core.check(core.lib.gnm_sum_partials(part.data_ptr(), part.numel(), out.data_ptr(), _stream()), "sum")
        want = float((dp64 * x32).sum())
        scale = float(np.abs(dp64 * x32).sum())
        assert abs(out.item() - want) <= 1e-6 * scale        # fp64 accumulation: far below fp32 noise


# (sizes, density, F): the 64-wide single-slice shape (gnm_agg16_kernel), and 32-float slices (gnm_agg_kernel<8, MODE>:
# hidden_dim 128 on graphs of 625-1231 nodes = configs[3]; 96 = three slices; 32 = one slice, small workgroups)
FUSED_AGG_CASES = [([40, 40, 40], 0.3, 64), ([400, 400], 0.3, 64), ([64] * 9, 0.5, 64), ([3] * 50, 0.9, 64),
                   ([9, 2, 12], 0.5, 64), ([700, 1000, 650], 0.02, 128), ([1100, 30, 1], 0.02, 96),
                   ([40, 9, 64, 1, 333], 0.3, 32), ([1000] * 3, 0.09, 128)]


@pytest.mark.parametrize("sizes,density,F", FUSED_AGG_CASES)
@pytest.mark.parametrize("average,learn_eps,graph_avg,disc", [(0, 1, 0, True), (1, 1, 1, True), (0, 0, 0, False),
                                                              (1, 0, 1, True)])
def test_agg_backward_fused_with_bn_stats(sizes, density, F, average, learn_eps, graph_avg, disc):
    """gnm_agg_bwd_stats == gnm_agg(backward) followed by gnm_bn_relu_bwd_stats (readout, discriminator and
    quirk-row terms, ReLU mask, per-graph BatchNorm-backward sums), to fp32 rounding."""
    from gnm import core
    from gnm._cabi import check, lib
    from gnm.arena import GraphArena
    rng = np.random.default_rng(len(sizes) * 1000 + average * 10 + learn_eps)
    graphs = random_graphs(rng, sizes, density, True)
    ar = GraphArena(DEV)
    batch = ar.batch(graphs)
    batch.dense = False              # CSR kernels on both sides of the comparison
    N, B = batch.N, batch.B
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dp = t(rng.standard_normal((N, F)).astype(np.float32))
    hf = t(rng.standard_normal((N, F)).astype(np.float32))
    Z = t((rng.standard_normal((N, F)) + 0.3).astype(np.float32))
    sc, sh = t(rng.uniform(0.5, 1.5, F).astype(np.float32)), t((rng.standard_normal(F) * 0.3).astype(np.float32))
    mu, rs = t(rng.standard_normal(F).astype(np.float32) * 0.1), t(rng.uniform(0.5, 1.5, F).astype(np.float32))
    dpool = t(rng.standard_normal((B, F)).astype(np.float32))
    dsc1 = t(rng.standard_normal(N).astype(np.float32)) if disc else None
    U = t(rng.standard_normal((B, 3 * F)).astype(np.float32))          # a [B, L*H] buffer; layer slice 1
    Ul = U[:, F:2 * F]
    perm = rng.permutation(B)
    inv = np.empty(B, dtype=np.int32); inv[perm] = np.arange(B, dtype=np.int32)
    invd, s2 = t(inv), t(rng.standard_normal(B).astype(np.float32))
    eps = torch.tensor([0.25], device=DEV)
    spec = core.GinSpec(1, 1, bool(learn_eps), "average" if graph_avg else "sum", "average" if average else "sum")
    epsp = eps.data_ptr() if learn_eps else None
    # reference: two kernels
    dh = torch.empty((N, F), device=DEV)
    part_e = torch.empty(lib.gnm_agg_num_partials(F, batch.n_max, B), dtype=torch.float64, device=DEV)
    core._agg(batch, dp, dh, F, epsp, spec, backward=True, hfwd=hf if learn_eps else None,
              deps_partial=part_e if learn_eps else None)
    G_ref = torch.empty((N, F), device=DEV)
    p_ref = torch.empty((B, 2, F), dtype=torch.float64, device=DEV)
    check(lib.gnm_bn_relu_bwd_stats(dh.data_ptr(), F, dpool.data_ptr(), F, graph_avg,
                                    dsc1.data_ptr() if disc else None, Ul.data_ptr() if disc else None,
                                    U.stride(0) if disc else 0, invd.data_ptr() if disc else None,
                                    s2.data_ptr() if disc else None, Z.data_ptr(), F, sc.data_ptr(), sh.data_ptr(),
                                    mu.data_ptr(), rs.data_ptr(), 1, G_ref.data_ptr(), F, batch.node_off.data_ptr(), B,
                                    F, p_ref.data_ptr(), _stream()), "stats")
    # fused
    G = torch.full((N, F), float("nan"), device=DEV)
    p_f = torch.full((B, 2, F), float("nan"), dtype=torch.float64, device=DEV)
    part_f = torch.empty_like(part_e)
    a = batch.arena
    check(lib.gnm_agg_bwd_stats(a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(),
                                batch.t_col_off.data_ptr(), a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(),
                                batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max, dp.data_ptr(), F,
                                G.data_ptr(), F, F, epsp, average, int(not learn_eps),
                                hf.data_ptr() if learn_eps else None, F if learn_eps else 0,
                                part_f.data_ptr() if learn_eps else None, Z.data_ptr(), F, sc.data_ptr(),
                                sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), dpool.data_ptr(), F, graph_avg,
                                dsc1.data_ptr() if disc else None, Ul.data_ptr() if disc else None,
                                U.stride(0) if disc else 0, invd.data_ptr() if disc else None,
                                s2.data_ptr() if disc else None, p_f.data_ptr(), _stream()), "agg_bwd_stats")
    assert_close(G.cpu().numpy(), G_ref.cpu().numpy(), rtol=TOL, what="G")
    fl = 1e-3 * float(G_ref.abs().sum(0).max())
    assert_close(p_f.cpu().numpy(), p_ref.cpu().numpy(), rtol=TOL, what="BN-backward partial sums", floor=fl)
    if learn_eps:
        assert torch.equal(part_f, part_e)                   # the d-eps partials are untouched by the fusion
        # hfwd = NULL: the layer input is recomputed in the epilogue as relu(Z*scale+shift); G and the BatchNorm
        # sums must not change and the partials must add up to sum dpooled . h in fp64
        G2 = torch.full((N, F), float("nan"), device=DEV)
        p2 = torch.full((B, 2, F), float("nan"), dtype=torch.float64, device=DEV)
        part2 = torch.full_like(part_e, float("nan"))
        check(lib.gnm_agg_bwd_stats(a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(),
                                    batch.t_col_off.data_ptr(), a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(),
                                    batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max, dp.data_ptr(), F,
                                    G2.data_ptr(), F, F, epsp, average, 0, None, 0, part2.data_ptr(), Z.data_ptr(), F,
                                    sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), dpool.data_ptr(), F,
                                    graph_avg, dsc1.data_ptr() if disc else None, Ul.data_ptr() if disc else None,
                                    U.stride(0) if disc else 0, invd.data_ptr() if disc else None,
                                    s2.data_ptr() if disc else None, p2.data_ptr(), _stream()), "agg_bwd_stats (h from Z)")
        assert torch.equal(G2, G) and torch.equal(p2, p_f)
        h_re = np.maximum(Z.cpu().numpy() * sc.cpu().numpy() + sh.cpu().numpy(), np.float32(0)).astype(np.float64)
        want = float((dp.cpu().numpy().astype(np.float64) * h_re).sum())
        scale_ = float(np.abs(dp.cpu().numpy().astype(np.float64) * h_re).sum())
        assert abs(float(part2.sum().item()) - want) <= 2e-6 * scale_
    # shapes it does not cover are refused
    assert lib.gnm_agg_bwd_stats(a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(),
                                 batch.t_col_off.data_ptr(), a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(),
                                 batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max, dp.data_ptr(), F,
                                 G.data_ptr(), F, 48, epsp, average, int(not learn_eps), None, 0, None, Z.data_ptr(), F,
                                 sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), None, 0, 0, None, None, 0,
                                 None, None, p_f.data_ptr(), _stream()) == -2


LIN_CASES = [(1000, 64, 64), (1000, 7, 64), (33, 64, 64), (1, 5, 32), (777, 128, 128), (500, 32, 32),
             (640, 400, 64), (300, 64, 7), (300, 64, 128), (257, 16, 96), (4096, 64, 64), (1, 128, 128), (40000 + 19, 128, 128)]


@pytest.mark.parametrize("N,K,H", LIN_CASES)
@pytest.mark.parametrize("pro,w_off", [(False, 0), (True, 0), (True, 5)])
def test_linear_forward_stats_and_grads(N, K, H, pro, w_off):
    """w_off: the weight sits w_off floats into a larger buffer, as a parameter inside the flat buffer of
    gnm/parallel.py does -- not 16-byte aligned (the streaming kernel then stages it with 4-byte loads)."""
    from gnm import core
    from gnm._cabi import lib
    rng = np.random.default_rng(N * 131 + K * 7 + H)
    X = rng.standard_normal((N, K)).astype(np.float32)
    W = (rng.standard_normal((H, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(H).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
    sh = rng.standard_normal(K).astype(np.float32)
    Xd, bd = (torch.from_numpy(a).to(DEV) for a in (X, b))
    flat = torch.zeros(w_off + H * K, device=DEV)
    Wd = flat[w_off:].view(H, K)
    Wd.copy_(torch.from_numpy(W))
    assert (Wd.data_ptr() % 16 == 0) == (w_off % 4 == 0)
    scd, shd = torch.from_numpy(sc).to(DEV), torch.from_numpy(sh).to(DEV)
    Z = torch.full((N, H), float("nan"), device=DEV)
    grid = lib.gnm_linear_grid(N)
    stats = torch.zeros((grid, 2, H), dtype=torch.float64, device=DEV)
    core._linear(Xd, Wd, 0, bd, Z, N, K, H, (scd, shd) if pro else None, stats)
    Xe = X.astype(np.float64)
    if pro:
        Xe = np.maximum(X * sc + sh, 0).astype(np.float64)  # fused BN-affine + ReLU prologue (mlp.py:48), fp32 like the kernel
    ref = Xe @ W.astype(np.float64).T + b                    # nn.Linear (mlp.py:43,49)
    assert_close(Z.cpu().numpy(), ref, rtol=TOL, what="linear fwd")
    st = stats.sum(0).cpu().numpy()
    z32 = Z.cpu().numpy().astype(np.float64)
    assert_close(st[0], z32.sum(0), rtol=1e-6, what="column sums")
    assert_close(st[1], (z32 ** 2).sum(0), rtol=1e-6, what="column sums of squares")
    # dX = dZ W  (same kernel, k-major weight)
    dZ = rng.standard_normal((N, H)).astype(np.float32)
    dZd = torch.from_numpy(dZ).to(DEV)
    dX = torch.full((N, K), float("nan"), device=DEV)
    core._linear_wide(dZd, Wd, 1, None, dX, N, H, K, None, None)
    assert_close(dX.cpu().numpy(), dZ.astype(np.float64) @ W.astype(np.float64), rtol=TOL, what="dgrad")
    # dW = dZ^T f(X), db = sum dZ
    dW = torch.full((H, K), float("nan"), device=DEV)
    db = torch.full((H,), float("nan"), device=DEV)
    ws = torch.empty(int(lib.gnm_wgrad_workspace_floats(N, H, K)), device=DEV)
    core.check(lib.gnm_linear_wgrad(dZd.data_ptr(), H, Xd.data_ptr(), K, N, H, K, scd.data_ptr() if pro else None,
                                    shd.data_ptr() if pro else None, 1 if pro else 0, dW.data_ptr(), K, db.data_ptr(),
                                    ws.data_ptr(), _stream()), "wgrad")
    assert_close(dW.cpu().numpy(), dZ.astype(np.float64).T @ Xe, rtol=TOL, what="wgrad")
    assert_close(db.cpu().numpy(), dZ.astype(np.float64).sum(0), rtol=TOL, what="bias grad")


@pytest.mark.parametrize("wide", [False, True])
def test_split_precision_products_k128_are_fp32_accurate(wide):
    """The K = H = 128 Linear of BASELINE configs[3] (csrc/linear.hip gnm_lin_split128_kernel): forward with the fused
    BatchNorm + ReLU prologue and the k-major (dX) form, element by element below 1e-6 x sum_k |x_k||w_k| against fp64."""
    from gnm import core
    N, K, H = 8192 + 45, 128, 128
    rng = np.random.default_rng(22)
    scale = (lambda shape: np.exp2(rng.integers(-12, 13, shape)).astype(np.float32)) if wide else (lambda shape: np.float32(1))
    X = (rng.standard_normal((N, K)) * scale((N, K))).astype(np.float32)
    W = (rng.standard_normal((H, K)) / 11 * scale((H, K))).astype(np.float32)
    b = rng.standard_normal(H).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32)
    sh = rng.standard_normal(K).astype(np.float32)
    Xd, Wd, bd, scd, shd = (torch.from_numpy(a).to(DEV) for a in (X, W, b, sc, sh))
    W64 = W.astype(np.float64)
    for pro in (None, (scd, shd)):
        Z = torch.empty(N, H, device=DEV)
        core._linear(Xd, Wd, 0, bd, Z, N, K, H, pro, None)
        X64 = (np.maximum(X * sc + sh, 0) if pro else X).astype(np.float64)
        err = np.abs(Z.cpu().numpy() - (X64 @ W64.T + b)) / (np.abs(X64) @ np.abs(W64).T + np.abs(b))
        print("K = 128 split-precision forward (prologue %s): max error %.2e of sum |x||w|" % (pro is not None, err.max()))
        # (fp32 accumulation of 128 exact products: its own worst case is 128 x 2^-24 = 7.6e-6; the wide-range case
        #  measures 0.9-1.1e-6 depending on the order the k steps are taken in)
        assert err.max() < 1.5e-6
    dX = torch.empty(N, K, device=DEV)
    core._linear(Xd, Wd, 1, None, dX, N, H, K, None, None)
    X64 = X.astype(np.float64)
    e2 = np.abs(dX.cpu().numpy() - X64 @ W64) / (np.abs(X64) @ np.abs(W64))
    print("K = 128 split-precision k-major form: max error %.2e" % e2.max())
    assert e2.max() < 1.5e-6
    # dW = dZ^T relu(sc X + sh) (gnm_wgrad_split128_kernel: batch row as the contraction index, 8,237 rows accumulated
    # in fp32 per workgroup and the workgroups' partials in a fixed order)
    from gnm._cabi import check, lib
    G = (rng.standard_normal((N, H)) * scale((N, H))).astype(np.float32)
    Gd = torch.from_numpy(G).to(DEV)
    dW, db = torch.empty(H, K, device=DEV), torch.empty(H, device=DEV)
    ws = torch.empty(int(lib.gnm_wgrad_workspace_floats(N, H, K)), device=DEV)
    for pro in (False, True):
        check(lib.gnm_linear_wgrad(Gd.data_ptr(), H, Xd.data_ptr(), K, N, H, K, scd.data_ptr() if pro else None,
                                   shd.data_ptr() if pro else None, 1 if pro else 0, dW.data_ptr(), K, db.data_ptr(),
                                   ws.data_ptr(), _stream()), "wgrad")
        X64 = (np.maximum(X * sc + sh, 0) if pro else X).astype(np.float64)
        G64 = G.astype(np.float64)
        e3 = np.abs(dW.cpu().numpy() - G64.T @ X64) / (np.abs(G64).T @ np.abs(X64))
        print("K = 128 split-precision wgrad (prologue %s): max error %.2e of sum |g||x|" % (pro, e3.max()))
        assert e3.max() < 2e-6
        assert_close(db.cpu().numpy(), G64.sum(0), rtol=TOL, what="bias grad")


@pytest.mark.parametrize("wide", [False, True])
def test_split_precision_products_are_fp32_accurate(wide):
    """K = H = 64 Linears run on the bf16 matrix pipe with both operands split into three exact bf16 planes and six of
    the nine partial products kept (csrc/linear.hip gnm_lin_split_kernel, SPLITD backward forms).  The claim is that
    this is an fp32-accurate product: element by element the error against fp64 must stay below 1e-6 x sum_k |x_k||w_k|
    -- a sequential fp32 dot product of length 64 is only guaranteed 64 x 6e-8 = 3.8e-6 of it -- also for operands
    whose magnitudes span many binades (`wide`: the dropped terms are relative to each product, not to the largest)."""
    from gnm import core
    from gnm._cabi import check, lib
    N, K, H = 4096 + 17, 64, 64
    rng = np.random.default_rng(21)
    scale = (lambda shape: np.exp2(rng.integers(-12, 13, shape)).astype(np.float32)) if wide else (lambda shape: np.float32(1))
    X = (rng.standard_normal((N, K)) * scale((N, K))).astype(np.float32)
    W = (rng.standard_normal((H, K)) / 8 * scale((H, K))).astype(np.float32)
    b = rng.standard_normal(H).astype(np.float32)
    Xd, Wd, bd = (torch.from_numpy(a).to(DEV) for a in (X, W, b))
    Z = torch.empty(N, H, device=DEV)
    core._linear(Xd, Wd, 0, bd, Z, N, K, H, None, None)
    X64, W64 = X.astype(np.float64), W.astype(np.float64)
    ref = X64 @ W64.T + b
    bound = np.abs(X64) @ np.abs(W64).T + np.abs(b)
    err = np.abs(Z.cpu().numpy() - ref) / bound
    print("split-precision forward: max error %.2e of sum |x||w| (wide = %s)" % (err.max(), wide))
    assert err.max() < 1e-6
    # dX of the fused backward (no-statistics form: BatchNorm-backward coefficients chosen so that dZ = G exactly)
    G = (rng.standard_normal((N, H)) * scale((N, H))).astype(np.float32)
    Gd = torch.from_numpy(G).to(DEV)
    zeros, ones = torch.zeros(H, device=DEV), torch.ones(H, device=DEV)
    dA, dW, db = torch.empty(N, K, device=DEV), torch.empty(H, K, device=DEV), torch.empty(H, device=DEV)
    ws = torch.empty(int(lib.gnm_linear_bwd_workspace_floats(N, H, K)), device=DEV)
    check(lib.gnm_linear_bwd_fused(Gd.data_ptr(), H, Gd.data_ptr(), H, zeros.data_ptr(), ones.data_ptr(), ones.data_ptr(),
                                   zeros.data_ptr(), zeros.data_ptr(), Xd.data_ptr(), K, None, None, 0, Wd.data_ptr(), K,
                                   dA.data_ptr(), K, dW.data_ptr(), K, db.data_ptr(), ws.data_ptr(), N, K, H, None, 0, None,
                                   None, None, None, None, _stream()), "linear_bwd_fused")
    G64 = G.astype(np.float64)
    e2 = np.abs(dA.cpu().numpy() - G64 @ W64) / (np.abs(G64) @ np.abs(W64))
    print("split-precision dgrad: max error %.2e of sum |g||w|" % e2.max())
    assert e2.max() < 1e-6
    e3 = np.abs(dW.cpu().numpy() - G64.T @ X64) / (np.abs(G64).T @ np.abs(X64))
    assert e3.max() < 2e-5            # (wgrad: fp32 instruction, 4,113 rows accumulated in fp32 partials)


@pytest.mark.parametrize("N,K,H", [(1000, 64, 64), (33, 64, 64), (4099, 32, 64), (257, 64, 32), (700, 32, 32),
                                   (1000, 7, 64), (97, 5, 32), (4099, 31, 64), (40, 1, 64)])   # K < 32: narrow variant
@pytest.mark.parametrize("pro,want_dx", [(True, True), (False, True), (True, False)])
def test_linear_backward_fused(N, K, H, pro, want_dx):
    """gnm_linear_bwd_fused = BatchNorm-backward apply + dX + dW + db in one pass, vs the
    oracle's bn_bwd formula and plain fp64 products."""
    from gnm._cabi import check, lib
    rng = np.random.default_rng(N + 17 * K + H)
    G = rng.standard_normal((N, H)).astype(np.float32)
    Z = (rng.standard_normal((N, H)) * 1.5 + 0.5).astype(np.float32)
    X = rng.standard_normal((N, K)).astype(np.float32)
    W = (rng.standard_normal((H, K)) / np.sqrt(K)).astype(np.float32)
    mean, rstd = Z.mean(0).astype(np.float32), (1 / np.sqrt(Z.var(0) + 1e-5)).astype(np.float32)
    cA = rng.uniform(0.5, 1.5, H).astype(np.float32)
    m1 = (G.mean(0)).astype(np.float32)
    m2 = rng.standard_normal(H).astype(np.float32) * 0.1
    sc, sh = rng.uniform(0.5, 1.5, K).astype(np.float32), rng.standard_normal(K).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    Gd, Zd, Xd, Wd, md, rd, cd, m1d, m2d, scd, shd = map(t, (G, Z, X, W, mean, rstd, cA, m1, m2, sc, sh))
    dA = torch.full((N, K), float("nan"), device=DEV) if want_dx else None
    dW = torch.full((H, K), float("nan"), device=DEV)
    db = torch.full((H,), float("nan"), device=DEV)
    ws = torch.empty(int(lib.gnm_linear_bwd_workspace_floats(N, H, K)), device=DEV)
    check(lib.gnm_linear_bwd_fused(Gd.data_ptr(), H, Zd.data_ptr(), H, md.data_ptr(), rd.data_ptr(), cd.data_ptr(),
                                   m1d.data_ptr(), m2d.data_ptr(), Xd.data_ptr(), K, scd.data_ptr() if pro else None,
                                   shd.data_ptr() if pro else None, 1 if pro else 0, Wd.data_ptr(), K,
                                   dA.data_ptr() if want_dx else None, K, dW.data_ptr(), K, db.data_ptr(),
                                   ws.data_ptr(), N, K, H, None, 0, None, None, None, None, None, _stream()),
          "linear_bwd_fused")
    f64 = np.float64
    xhat = (Z.astype(f64) - mean) * rstd
    dZ = cA.astype(f64) * (G.astype(f64) - m1 - xhat * m2)
    Xe = np.maximum(X * sc + sh, 0).astype(f64) if pro else X.astype(f64)
    if want_dx:
        assert_close(dA.cpu().numpy(), dZ @ W.astype(f64), rtol=TOL, what="dX")
    assert_close(dW.cpu().numpy(), dZ.T @ Xe, rtol=TOL, what="dW")
    # db is a cancelling sum here (G - mean(G) sums to ~0): bound it by fp32 eps x sum |dZ|
    assert_close(db.cpu().numpy(), dZ.sum(0), rtol=TOL, what="db", floor=1e-2 * np.abs(dZ).sum(0).max())
    # ineligible shapes are refused, not mis-computed
    assert lib.gnm_linear_bwd_fused(Gd.data_ptr(), H, Zd.data_ptr(), H, md.data_ptr(), rd.data_ptr(), cd.data_ptr(),
                                    m1d.data_ptr(), m2d.data_ptr(), Xd.data_ptr(), K, None, None, 0, Wd.data_ptr(), K,
                                    None, K, dW.data_ptr(), K, db.data_ptr(), ws.data_ptr(), N, 40, H, None, 0, None,
                                    None, None, None, None, _stream()) == -2
    if want_dx and K < 32:
        # the narrow variant has no lower-BatchNorm epilogue (an input layer has nothing below it)
        assert lib.gnm_linear_bwd_fused(Gd.data_ptr(), H, Zd.data_ptr(), H, md.data_ptr(), rd.data_ptr(),
                                        cd.data_ptr(), m1d.data_ptr(), m2d.data_ptr(), Xd.data_ptr(), K, None, None, 0,
                                        Wd.data_ptr(), K, dA.data_ptr(), K, dW.data_ptr(), K, db.data_ptr(),
                                        ws.data_ptr(), N, K, H, Gd.data_ptr(), H, md.data_ptr(), md.data_ptr(),
                                        md.data_ptr(), md.data_ptr(), ws.data_ptr(), _stream()) == -2
    if want_dx and K >= 32:
        # variant that also applies the ReLU mask of the BatchNorm+ReLU feeding this Linear to dX and
        # reduces that BatchNorm's backward sums (what gnm_bn_relu_bwd_stats would do)
        Zlo = (rng.standard_normal((N, K)) + 0.2).astype(np.float32)
        lsc, lsh = rng.uniform(0.5, 1.5, K).astype(np.float32), (rng.standard_normal(K) * 0.3).astype(np.float32)
        lmu, lrs = Zlo.mean(0).astype(np.float32), (1 / np.sqrt(Zlo.var(0) + 1e-5)).astype(np.float32)
        Zld, lscd, lshd, lmud, lrsd = map(t, (Zlo, lsc, lsh, lmu, lrs))
        grid = lib.gnm_linear_bwd_grid(N)
        part = torch.full((grid, 2, K), float("nan"), dtype=torch.float64, device=DEV)
        dA2 = torch.full((N, K), float("nan"), device=DEV)
        check(lib.gnm_linear_bwd_fused(Gd.data_ptr(), H, Zd.data_ptr(), H, md.data_ptr(), rd.data_ptr(), cd.data_ptr(),
                                       m1d.data_ptr(), m2d.data_ptr(), Xd.data_ptr(), K,
                                       scd.data_ptr() if pro else None, shd.data_ptr() if pro else None,
                                       1 if pro else 0, Wd.data_ptr(), K, dA2.data_ptr(), K, dW.data_ptr(), K,
                                       db.data_ptr(), ws.data_ptr(), N, K, H, Zld.data_ptr(), K, lscd.data_ptr(),
                                       lshd.data_ptr(), lmud.data_ptr(), lrsd.data_ptr(), part.data_ptr(), _stream()),
              "linear_bwd_fused + stats")
        mask = (Zlo * lsc + lsh) > 0                           # fp32, as the kernel evaluates it
        Gref = (dZ @ W.astype(f64)) * mask
        xh = (Zlo.astype(f64) - lmu) * lrs
        assert_close(dA2.cpu().numpy(), Gref, rtol=TOL, what="masked dX")
        st = part.sum(0).cpu().numpy()
        fl = 1e-2 * np.abs(Gref).sum(0).max()
        assert_close(st[0], Gref.sum(0), rtol=TOL, what="sum g", floor=fl)
        assert_close(st[1], (Gref * xh).sum(0), rtol=TOL, what="sum g*xhat", floor=fl)
        assert_close(dW.cpu().numpy(), dZ.T @ Xe, rtol=TOL, what="dW (stats variant)")
        if pro:
            # the shape the model has for the second Linear of an MLP: the lower BatchNorm's input IS X and its affine
            # IS the prologue -- mask and sums then come from the X registers of the weight-gradient product (K = H =
            # 64 takes that variant; the others must give the same through the general one)
            lmu2, lrs2 = X.mean(0).astype(np.float32), (1 / np.sqrt(X.var(0) + 1e-5)).astype(np.float32)
            lmu2d, lrs2d = t(lmu2), t(lrs2)
            part2 = torch.full((grid, 2, K), float("nan"), dtype=torch.float64, device=DEV)
            dA3 = torch.full((N, K), float("nan"), device=DEV)
            dW3 = torch.full((H, K), float("nan"), device=DEV)
            check(lib.gnm_linear_bwd_fused(Gd.data_ptr(), H, Zd.data_ptr(), H, md.data_ptr(), rd.data_ptr(),
                                           cd.data_ptr(), m1d.data_ptr(), m2d.data_ptr(), Xd.data_ptr(), K,
                                           scd.data_ptr(), shd.data_ptr(), 1, Wd.data_ptr(), K, dA3.data_ptr(), K,
                                           dW3.data_ptr(), K, db.data_ptr(), ws.data_ptr(), N, K, H, Xd.data_ptr(), K,
                                           scd.data_ptr(), shd.data_ptr(), lmu2d.data_ptr(), lrs2d.data_ptr(),
                                           part2.data_ptr(), _stream()), "linear_bwd_fused + stats (sZ = X)")
            mask2 = (X * sc + sh) > 0
            Gref2 = (dZ @ W.astype(f64)) * mask2
            xh2 = (X.astype(f64) - lmu2) * lrs2
            assert_close(dA3.cpu().numpy(), Gref2, rtol=TOL, what="masked dX (sZ = X)")
            st2 = part2.sum(0).cpu().numpy()
            fl2 = 1e-2 * np.abs(Gref2).sum(0).max()
            assert_close(st2[0], Gref2.sum(0), rtol=TOL, what="sum g (sZ = X)", floor=fl2)
            assert_close(st2[1], (Gref2 * xh2).sum(0), rtol=TOL, what="sum g*xhat (sZ = X)", floor=fl2)
            assert_close(dW3.cpu().numpy(), dZ.T @ Xe, rtol=TOL, what="dW (sZ = X)")


@pytest.mark.parametrize("wg16", ["1", "0"])
def test_linear_backward_recomputing_its_output(wg16):
    """gnm_linear_bwd_fused_rz (K = H = 64): the pass that recomputes Z = f(X) W^T + b instead of reading it, against
    fp64 and against gnm_linear_bwd_fused fed the Z that gnm_linear_fwd wrote.  samez: the second Linear of an MLP
    (prologue = the lower BatchNorm + ReLU; its mask and backward sums come out of the same pass).  wg16: the weight
    gradient on the bf16 pipe / with the fp32 instruction (a process-wide knob: run in a child process)."""
    code = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from gnm import core
from gnm._cabi import check, lib
DEV = torch.device("cuda:0")
K = H = 64
for N, samez in [(n, z) for n in (1, 31, 4096 + 77, 40000 + 5) for z in (0, 1)] + [(409600, 0)]:     # (+ BASELINE's full size)
  rng = np.random.default_rng(N + samez)
  t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(DEV)
  X = (rng.standard_normal((N, K)) * 1.5 + 0.3).astype(np.float32)
  W = (rng.standard_normal((H, K)) / 8).astype(np.float32)
  b = rng.standard_normal(H).astype(np.float32)
  G = rng.standard_normal((N, H)).astype(np.float32)
  sc, sh = rng.uniform(0.5, 1.5, K).astype(np.float32), rng.standard_normal(K).astype(np.float32)
  lmu, lrs = rng.standard_normal(K).astype(np.float32), rng.uniform(0.5, 1.5, K).astype(np.float32)
  mean, rstd, cA, m1, m2 = (rng.uniform(0.5, 1.5, H).astype(np.float32) for _ in range(5))
  Xd, Wd, bd, Gd, scd, shd, lmud, lrsd = map(t, (X, W, b, G, sc, sh, lmu, lrs))
  vec = [t(v) for v in (mean, rstd, cA, m1, m2)]
  st = torch.cuda.current_stream().cuda_stream
  Z = torch.empty(N, H, device=DEV)
  core._linear(Xd, Wd, 0, bd, Z, N, K, H, (scd, shd) if samez else None, None)
  grid = lib.gnm_linear_bwd_grid(N)
  out = []
  for rz in (True, False):
      dA = torch.full((N, K), float("nan"), device=DEV)
      dW = torch.full((H, K), float("nan"), device=DEV)
      db = torch.full((H,), float("nan"), device=DEV)
      ws = torch.empty(int(lib.gnm_linear_bwd_workspace_floats(N, H, K)), device=DEV)
      part = torch.full((grid, 2, K), float("nan"), dtype=torch.float64, device=DEV)
      lo = (Xd.data_ptr(), K, scd.data_ptr(), shd.data_ptr(), lmud.data_ptr(), lrsd.data_ptr(), part.data_ptr()) if samez \
          else (None, 0, None, None, None, None, None)
      pro = (scd.data_ptr(), shd.data_ptr(), 1) if samez else (None, None, 0)
      tail = (Xd.data_ptr(), K) + pro + (Wd.data_ptr(), K, dA.data_ptr(), K, dW.data_ptr(), K, db.data_ptr(), ws.data_ptr(),
                                        N, K, H) + lo + (st,)
      if rz:
          check(lib.gnm_linear_bwd_fused_rz(Gd.data_ptr(), H, bd.data_ptr(), *[v.data_ptr() for v in vec], *tail), "rz")
      else:
          check(lib.gnm_linear_bwd_fused(Gd.data_ptr(), H, Z.data_ptr(), H, *[v.data_ptr() for v in vec], *tail), "fused")
      torch.cuda.synchronize()
      out.append([a.cpu().numpy() for a in (dA, dW, db, part.sum(0))])
  # fp64 reference
  X64 = X.astype(np.float64)
  F64 = np.maximum(X64 * sc + sh, 0) if samez else X64
  Z64 = F64 @ W.astype(np.float64).T + b
  dZ = cA * (G - m1 - (Z64 - mean) * rstd * m2)
  dA_ref = dZ @ W.astype(np.float64)
  if samez:
      dA_ref = dA_ref * ((X * sc + sh) > 0)
  def close(got, ref, tol, what):
      err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)          # max-norm relative, as everywhere in this file
      assert err < tol, (what, err)
  for name, (dA, dW, db, ps) in zip(("rz", "stored Z"), out):
      close(dA, dA_ref, 1e-5, name + " dA")
      close(dW, dZ.T @ F64, 1e-5, name + " dW")
      close(db, dZ.sum(0), 1e-5, name + " db")
      if samez:
          close(ps[0], dA_ref.sum(0), 1e-5, name + " sum g")
          close(ps[1], (dA_ref * ((X64 - lmu) * lrs)).sum(0), 1e-5, name + " sum g xhat")
  # the two passes see the same Z (the recomputation repeats gnm_lin_split_kernel's instruction sequence)
  dmax = np.abs(out[0][0] - out[1][0]).max()
  print("N", N, "samez", samez, "max |dA(rz) - dA(stored Z)| =", dmax)
  assert dmax <= 1e-6 * np.abs(out[1][0]).max()
print("ok")
""" % (ROOT, os.path.join(ROOT, "graph-neural-mapping_amd"))
    env = dict(os.environ, GNM_LINBWD_WG16=wg16, GNM_LINBWD_RZ_STATS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_linear_backward_deferred_weight_reduction():
    """dW = NULL leaves the per-workgroup partials in the workspace; gnm_reduce_partials_multi reduces several
    calls' partials in one launch and must give the bits of the immediate reduction."""
    import ctypes as C
    from gnm._cabi import check, lib
    N = 5000
    rng = np.random.default_rng(5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(DEV)
    jobs = []
    for K, H in [(64, 64), (7, 64), (32, 64), (64, 32)]:
        G, Z, X = t(rng.standard_normal((N, H))), t(rng.standard_normal((N, H))), t(rng.standard_normal((N, K)))
        W = t(rng.standard_normal((H, K)) / np.sqrt(K))
        vec = [t(rng.uniform(0.5, 1.5, H)) for _ in range(5)]
        got = []
        for defer in (False, True):
            dW = torch.full((H, K + 3), float("nan"), device=DEV)          # a row stride that is not K
            db = torch.full((H,), float("nan"), device=DEV)
            ws = torch.empty(int(lib.gnm_linear_bwd_workspace_floats(N, H, K)), device=DEV)
            dA = torch.empty(N, K, device=DEV)
            check(lib.gnm_linear_bwd_fused(G.data_ptr(), H, Z.data_ptr(), H, *[v.data_ptr() for v in vec], X.data_ptr(), K,
                                           None, None, 0, W.data_ptr(), K, dA.data_ptr(), K,
                                           None if defer else dW.data_ptr(), K + 3, db.data_ptr(), ws.data_ptr(), N, K, H,
                                           None, 0, None, None, None, None, None, _stream()), "linear_bwd_fused")
            got.append((dW, db, ws))
        assert torch.isnan(got[1][0]).all() and torch.isnan(got[1][1]).all()       # deferred: nothing written yet
        jobs.append((got[0], got[1], H, K))
    nj = len(jobs)
    check(lib.gnm_reduce_partials_multi(
        (C.c_void_p * nj)(*[j[1][2].data_ptr() for j in jobs]), (C.c_void_p * nj)(*[j[1][0].data_ptr() for j in jobs]),
        (C.c_int * nj)(*[j[3] + 3 for j in jobs]), (C.c_void_p * nj)(*[j[1][1].data_ptr() for j in jobs]),
        (C.c_int * nj)(*[j[2] for j in jobs]), (C.c_int * nj)(*[j[3] for j in jobs]), nj, N, _stream()), "reduce_multi")
    for now, later, H, K in jobs:
        assert torch.equal(now[0][:, :K], later[0][:, :K]) and torch.equal(now[1], later[1])
        assert torch.isnan(later[0][:, K:]).all()
    assert lib.gnm_reduce_partials_multi(None, None, None, None, None, None, 33, N, _stream()) != 0


@pytest.mark.parametrize("sizes,H", [([40, 40, 40], 64), ([13, 50, 7], 32), ([300, 300], 128), ([1], 64)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("average", [0, 1])
def test_batchnorm_relu_readout_forward_backward(sizes, H, training, average):
    """gnm_bn_finalize / gnm_bn_relu_readout / gnm_bn_relu_bwd_stats / _finalize / _apply vs
    the oracle's bn_fwd / bn_bwd (torch.nn.BatchNorm1d semantics) and the readout spmm."""
    from gnm._cabi import check, lib
    from oracle import gin_oracle as O
    rng = np.random.default_rng(sum(sizes) + H)
    N, B = sum(sizes), len(sizes)
    node_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    z = (rng.standard_normal((N, H)) * 2 + 3).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, H).astype(np.float32)
    beta = rng.standard_normal(H).astype(np.float32)
    rm = rng.standard_normal(H).astype(np.float32)
    rv = rng.uniform(0.5, 2, H).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    zd, gd, bd, rmd, rvd, nod = t(z), t(gamma), t(beta), t(rm.copy()), t(rv.copy()), t(node_off)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    # statistics partials as the Linear epilogue would write them (2 blocks)
    half = N // 2
    parts = np.zeros((2, 2, H))
    for i, sl in enumerate((slice(0, half), slice(half, N))):
        parts[i, 0] = z[sl].astype(np.float64).sum(0)
        parts[i, 1] = (z[sl].astype(np.float64) ** 2).sum(0)
    pd = t(parts)
    scale, shift, mean, rstd = (torch.empty(H, device=DEV) for _ in range(4))
    check(lib.gnm_bn_finalize(pd.data_ptr(), 2, H, N, gd.data_ptr(), bd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(),
                              nbt.data_ptr(), 0.1, 1e-5, int(training), 1, scale.data_ptr(), shift.data_ptr(),
                              mean.data_ptr(), rstd.data_ptr(), _stream()), "bn_finalize")
    y, cache, (nrm, nrv) = O.bn_fwd(z.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64),
                                    rm.astype(np.float64), rv.astype(np.float64), training)
    if N > 1 or not training:
        assert_close(rmd.cpu().numpy(), nrm, rtol=TOL, what="running_mean")
        assert_close(rvd.cpu().numpy(), nrv, rtol=TOL, what="running_var")
    assert int(nbt.item()) == (1 if training else 0)
    hout = torch.full((N, H), float("nan"), device=DEV)
    pooled = torch.full((B, 2 * H), float("nan"), device=DEV)      # strided destination (slice of g_f)
    check(lib.gnm_bn_relu_readout(zd.data_ptr(), H, scale.data_ptr(), shift.data_ptr(), hout.data_ptr(), H,
                                  nod.data_ptr(), B, H, 1, pooled[:, H:].data_ptr(), 2 * H, average, _stream()),
          "bn_relu_readout")
    href = np.maximum(y, 0)
    if N > 1 or not training:
        assert_close(hout.cpu().numpy(), href, rtol=TOL, what="relu(bn(z))")
        pref = np.stack([href[node_off[b]:node_off[b + 1]].sum(0) * ((1.0 / sizes[b]) if average else 1.0)
                         for b in range(B)])
        assert_close(pooled[:, H:].cpu().numpy(), pref, rtol=TOL, what="readout")
    if N == 1 and training:
        return
    # backward
    dH = rng.standard_normal((N, H)).astype(np.float32)
    dpool = rng.standard_normal((B, H)).astype(np.float32)
    total = dH.astype(np.float64)
    for b in range(B):
        total[node_off[b]:node_off[b + 1]] += dpool[b] * ((1.0 / sizes[b]) if average else 1.0)
    dy = total * (y > 0)
    dx_ref, dg_ref, db_ref = O.bn_bwd(dy, cache)
    G = torch.empty((N, H), device=DEV)
    part = torch.empty((B, 2, H), dtype=torch.float64, device=DEV)
    dHd, dpoold = t(dH), t(dpool)                     # keep the device copies alive across the launches
    check(lib.gnm_bn_relu_bwd_stats(dHd.data_ptr(), H, dpoold.data_ptr(), H, average, None, None, 0, None, None,
                                    zd.data_ptr(), H, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                    rstd.data_ptr(), 1, G.data_ptr(), H, nod.data_ptr(), B, H, part.data_ptr(),
                                    _stream()), "bwd_stats")
    dgam, dbet, cA, m1, m2 = (torch.empty(H, device=DEV) for _ in range(5))
    check(lib.gnm_bn_bwd_finalize(part.data_ptr(), B, H, N, gd.data_ptr(), rstd.data_ptr(), int(training),
                                  dgam.data_ptr(), dbet.data_ptr(), cA.data_ptr(), m1.data_ptr(), m2.data_ptr(),
                                  _stream()), "bwd_finalize")
    check(lib.gnm_bn_bwd_apply(G.data_ptr(), H, zd.data_ptr(), H, mean.data_ptr(), rstd.data_ptr(), cA.data_ptr(),
                               m1.data_ptr(), m2.data_ptr(), G.data_ptr(), H, N, H, _stream()), "bwd_apply")
    floor = 1e-3 * float(np.abs(dy).max())
    assert_close(G.cpu().numpy(), dx_ref, rtol=TOL, what="BN backward dx", floor=floor)
    assert_close(dgam.cpu().numpy(), dg_ref, rtol=TOL, what="dgamma", floor=floor)
    assert_close(dbet.cpu().numpy(), db_ref, rtol=TOL, what="dbeta", floor=floor)


@pytest.mark.parametrize("B,n,L,H", [(3, 24, 5, 64), (4, 20, 3, 32), (2, 1, 2, 64), (5, 400, 5, 64), (3, 17, 3, 20)])
@pytest.mark.parametrize("as_z", [False, True])
def test_discriminator_scores_vs_literal_bilinear(B, n, L, H, as_z):
    """gnm_disc_score_fwd/_bwd vs the literal nn.Bilinear formulation of discriminator.py:19-38
    (oracle restatement), including the graph-index-as-row-index shuffle quirk.
    as_z: all layers but the last are handed over as the pre-BatchNorm Z with folded (scale, shift), the way the model
    does when the activation is never written; the kernels re-form h = relu(Z * scale + shift) (graphcnn.py:163-166)."""
    from gnm._cabi import check, lib
    rng = np.random.default_rng(B * 100 + n)
    N, LH = B * n, L * H
    zs = [rng.standard_normal((N, H)).astype(np.float32) for _ in range(L)]
    scs = [rng.uniform(0.5, 1.5, H).astype(np.float32) for _ in range(L)]
    shs = [(rng.standard_normal(H) * 0.3).astype(np.float32) for _ in range(L)]
    given_as_z = [as_z and l < L - 1 for l in range(L)]
    hs = [np.maximum(zs[l] * scs[l] + shs[l], np.float32(0)) if given_as_z[l] else zs[l] for l in range(L)]
    n_f = np.concatenate(hs, 1).astype(np.float64)
    c = rng.uniform(0, 1, (B, LH)).astype(np.float32)
    Wd = (rng.standard_normal((LH, LH)) / LH).astype(np.float32)
    bias = np.float32(0.3)
    perm = rng.permutation(B)
    idx = np.repeat(perm, n)                                  # graphcnn.py:198-201
    c_x = np.repeat(c.astype(np.float64), n, axis=0)
    sc1 = ((n_f @ Wd.astype(np.float64)) * c_x).sum(1) + bias
    sc2 = ((n_f[idx] @ Wd.astype(np.float64)) * c_x).sum(1) + bias
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    hd = [t(z) for z in zs]                                   # arrays as handed to the kernels: Z or h per layer
    scd, shd = [t(a) for a in scs], [t(a) for a in shs]
    U = t((c.astype(np.float64) @ Wd.astype(np.float64).T).astype(np.float32))
    node_off = t(np.arange(B + 1, dtype=np.int32) * n)
    perm_rows = t(perm.astype(np.int32))
    d_logit = torch.empty(2 * N, device=DEV)
    hp = (C.c_void_p * L)(*[h.data_ptr() for h in hd])
    sp = (C.c_void_p * L)(*[scd[l].data_ptr() if given_as_z[l] else None for l in range(L)]) if as_z else None
    tp = (C.c_void_p * L)(*[shd[l].data_ptr() if given_as_z[l] else None for l in range(L)]) if as_z else None
    bd = t(np.array([bias]))
    check(lib.gnm_disc_score_fwd(hp, sp, tp, H, L, H, U.data_ptr(), LH, perm_rows.data_ptr(), bd.data_ptr(),
                                 node_off.data_ptr(), N, B, d_logit.data_ptr(), _stream()), "disc fwd")
    assert_close(d_logit.cpu().numpy(), np.concatenate([sc1, sc2]), rtol=TOL, what="d_logit")
    # backward wrt U and the negative-branch sums
    dD = rng.standard_normal(2 * N).astype(np.float32)
    dU = torch.empty((B, LH), device=DEV)
    s2 = torch.empty(B, device=DEV)
    dDd = t(dD)
    dsum = torch.empty(B, device=DEV)
    inv_perm = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    check(lib.gnm_disc_score_bwd(hp, sp, tp, H, L, H, dDd.data_ptr(), perm_rows.data_ptr(), node_off.data_ptr(), N, B,
                                 dU.data_ptr(), LH, s2.data_ptr(), dsum.data_ptr(), inv_perm.data_ptr(), _stream()),
          "disc bwd")
    d1, d2 = dD[:N].astype(np.float64), dD[N:].astype(np.float64)
    s2_ref = d2.reshape(B, n).sum(1)
    assert_close(dsum.cpu().numpy(), s2_ref + d1.reshape(B, n).sum(1), rtol=TOL, what="dsum",
                 floor=1e-3 * np.abs(dD).max())
    assert np.array_equal(inv_perm.cpu().numpy(), np.argsort(perm).astype(np.int32))      # bit-exact index structure
    dU_ref = np.stack([(d1[g * n:(g + 1) * n, None] * n_f[g * n:(g + 1) * n]).sum(0) + s2_ref[g] * n_f[perm[g]]
                       for g in range(B)])
    assert_close(s2.cpu().numpy(), s2_ref, rtol=TOL, what="s2sum", floor=1e-3 * np.abs(d2).max())
    assert_close(dU.cpu().numpy(), dU_ref, rtol=TOL, what="dU", floor=1e-3 * np.abs(dU_ref).max())


@pytest.mark.parametrize("B,n,L,H", [(3, 24, 5, 64), (4, 20, 3, 32), (2, 1, 2, 64), (5, 400, 5, 64), (1100, 8, 5, 64),
                                     (3, 40, 4, 128)])
@pytest.mark.parametrize("as_z", [False, True])
def test_discriminator_unit_reductions_equal_the_backward_pass(B, n, L, H, as_z):
    """gnm_disc_score_fwd_unit: same d_logit as gnm_disc_score_fwd (to fp32 rounding: the two instantiations of the
    kernel body contract their multiply-adds differently), and its by-products,
    scaled by gnm_disc_unit_scale, are what gnm_disc_score_bwd computes from dD = k (sigmoid(d_logit) - target) -- the
    gradient of beta * BCEWithLogits against ones / zeros (main.py:32-37) -- and what fp64 says.  (1100 graphs: the
    256-thread workgroup form; 3 graphs: 1024 threads.)"""
    from gnm._cabi import check, lib
    rng = np.random.default_rng(B * 7 + n)
    N, LH = B * n, L * H
    zs = [rng.standard_normal((N, H)).astype(np.float32) for _ in range(L)]
    scs = [rng.uniform(0.5, 1.5, H).astype(np.float32) for _ in range(L)]
    shs = [(rng.standard_normal(H) * 0.3).astype(np.float32) for _ in range(L)]
    given_as_z = [as_z and l < L - 1 for l in range(L)]
    hs = [np.maximum(zs[l] * scs[l] + shs[l], np.float32(0)) if given_as_z[l] else zs[l] for l in range(L)]
    n_f = np.concatenate(hs, 1).astype(np.float64)
    perm = rng.permutation(B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    hd = [t(z) for z in zs]
    scd, shd = [t(a) for a in scs], [t(a) for a in shs]
    U = t((rng.standard_normal((B, LH)) * 0.05).astype(np.float32))
    node_off = t(np.arange(B + 1, dtype=np.int32) * n)
    perm_rows = t(perm.astype(np.int32))
    hp = (C.c_void_p * L)(*[h.data_ptr() for h in hd])
    sp = (C.c_void_p * L)(*[scd[l].data_ptr() if given_as_z[l] else None for l in range(L)]) if as_z else None
    tp = (C.c_void_p * L)(*[shd[l].data_ptr() if given_as_z[l] else None for l in range(L)]) if as_z else None
    bd = t(np.array([0.3], dtype=np.float32))
    d_ref = torch.empty(2 * N, device=DEV)
    check(lib.gnm_disc_score_fwd(hp, sp, tp, H, L, H, U.data_ptr(), LH, perm_rows.data_ptr(), bd.data_ptr(),
                                 node_off.data_ptr(), N, B, d_ref.data_ptr(), _stream()), "disc fwd")
    ldunit = (LH + 2 + 3) & ~3
    unit = torch.full((B, ldunit), float("nan"), device=DEV)
    inv_perm = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    d_logit = torch.empty(2 * N, device=DEV)
    check(lib.gnm_disc_score_fwd_unit(hp, sp, tp, H, L, H, U.data_ptr(), LH, perm_rows.data_ptr(), bd.data_ptr(),
                                      node_off.data_ptr(), N, B, d_logit.data_ptr(), unit.data_ptr(), ldunit,
                                      inv_perm.data_ptr(), _stream()), "disc fwd unit")
    assert_close(d_logit.cpu().numpy(), d_ref.cpu().numpy().astype(np.float64), rtol=2e-6, what="d_logit (unit form)")
    assert np.array_equal(inv_perm.cpu().numpy(), np.argsort(perm).astype(np.int32))
    # the loss's gradient: dD = k (sigmoid(x) - target), k = upstream * beta / (2N)
    k = np.float32(1.7) * (np.float32(0.05) / np.float32(2 * N))
    x64 = d_ref.cpu().numpy().astype(np.float64)
    tgt = np.concatenate([np.ones(N), np.zeros(N)])
    dD64 = float(k) * (1.0 / (1.0 + np.exp(-x64)) - tgt)
    kd = t(np.array([k], dtype=np.float32))
    dU, s2, dsum = torch.empty((B, LH), device=DEV), torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    dbias = torch.full((1,), float("nan"), device=DEV)
    check(lib.gnm_disc_unit_scale(unit.data_ptr(), ldunit, LH, kd.data_ptr(), 1.0, B, dU.data_ptr(), LH, s2.data_ptr(),
                                  dsum.data_ptr(), dbias.data_ptr(), _stream()), "unit scale")
    # the Bilinear bias gradient of the same launch = the total of dsum (discriminator.py:19)
    assert abs(float(dbias.item()) - float(dD64.sum())) <= 1e-5 * float(np.abs(dD64).sum())
    # (a) against the fp64 statement of the reductions
    d1, d2 = dD64[:N], dD64[N:]
    s2_ref = d2.reshape(B, n).sum(1)
    dU_ref = np.stack([(d1[g * n:(g + 1) * n, None] * n_f[g * n:(g + 1) * n]).sum(0) + s2_ref[g] * n_f[perm[g]]
                       for g in range(B)])
    assert_close(s2.cpu().numpy(), s2_ref, rtol=TOL, what="s2sum")
    assert_close(dsum.cpu().numpy(), s2_ref + d1.reshape(B, n).sum(1), rtol=TOL, what="dsum",
                 floor=1e-3 * np.abs(dD64).max() * n)
    assert_close(dU.cpu().numpy(), dU_ref, rtol=TOL, what="dU", floor=1e-3 * np.abs(dU_ref).max())
    # (b) against the kernel pass it replaces, fed the fp32 gradient a loss kernel would write
    dDd = t(dD64.astype(np.float32))
    dU_b, s2_b, dsum_b = torch.empty_like(dU), torch.empty_like(s2), torch.empty_like(dsum)
    ip_b = torch.empty_like(inv_perm)
    check(lib.gnm_disc_score_bwd(hp, sp, tp, H, L, H, dDd.data_ptr(), perm_rows.data_ptr(), node_off.data_ptr(), N, B,
                                 dU_b.data_ptr(), LH, s2_b.data_ptr(), dsum_b.data_ptr(), ip_b.data_ptr(), _stream()),
          "disc bwd")
    assert torch.equal(ip_b, inv_perm)
    assert_close(dU.cpu().numpy(), dU_b.cpu().numpy(), rtol=TOL, what="dU vs pass", floor=1e-3 * np.abs(dU_ref).max())
    assert_close(s2.cpu().numpy(), s2_b.cpu().numpy(), rtol=TOL, what="s2sum vs pass")


def test_discriminator_unit_declines_outside_its_forms():
    from gnm._cabi import lib
    B, n, L, H = 2, 8, 3, 20                                    # H / 4 = 5: not a vector form
    z = [torch.randn(B * n, H, device=DEV) for _ in range(L)]
    hp = (C.c_void_p * L)(*[h.data_ptr() for h in z])
    U = torch.randn(B, L * H, device=DEV)
    node_off = torch.arange(B + 1, dtype=torch.int32, device=DEV) * n
    perm = torch.arange(B, dtype=torch.int32, device=DEV)
    d = torch.empty(2 * B * n, device=DEV)
    unit = torch.empty(B, 64, device=DEV)
    ip = torch.empty(B, dtype=torch.int32, device=DEV)
    rc = lib.gnm_disc_score_fwd_unit(hp, None, None, H, L, H, U.data_ptr(), L * H, perm.data_ptr(), None,
                                     node_off.data_ptr(), B * n, B, d.data_ptr(), unit.data_ptr(), 64, ip.data_ptr(),
                                     _stream())
    assert rc == -2


@pytest.mark.parametrize("sizes,density,F", [([40, 40, 40], 0.3, 64), ([37, 5, 64, 1, 23], 0.4, 64), ([400, 400, 400], 0.3, 64),
                                             ([50, 50], 0.0, 64), ([700, 1000, 650], 0.02, 128), ([1100, 30, 1], 0.02, 96),
                                             ([40, 9, 64, 1, 333], 0.3, 32), ([1000] * 3, 0.09, 128)])
@pytest.mark.parametrize("average,learn_eps,graph_avg", [(0, 1, 0), (1, 1, 1), (0, 0, 1), (1, 0, 0)])
def test_agg_forward_with_fused_bn_relu_readout(sizes, density, F, average, learn_eps, graph_avg):
    """gnm_agg_fwd_bnrelu (previous layer's BatchNorm + ReLU + readout on the aggregation's tile load) vs the fp64
    restatement of graphcnn.py:163-166, 229, 154-161 / 178-182."""
    from gnm._cabi import check, lib
    from gnm.arena import GraphArena
    rng = np.random.default_rng(len(sizes) * 1000 + sizes[0])
    graphs = random_graphs(rng, sizes, density, True)
    ar = GraphArena(DEV)
    batch = ar.batch(graphs)
    A = dense_adj(graphs)
    N, B = batch.N, batch.B
    z = rng.standard_normal((N, F)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, F).astype(np.float32)
    sh = (rng.standard_normal(F) * 0.3).astype(np.float32)
    eps = 0.37
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    zd, scd, shd, epsd = t(z), t(sc), t(sh), torch.tensor([eps], device=DEV)
    hout = torch.full((N, F), float("nan"), device=DEV)
    gf = torch.full((B, 3 * F), float("nan"), device=DEV)          # the readout lands in a column window of g_f
    y = torch.full((N, F), float("nan"), device=DEV)
    gwin = gf[:, F:2 * F]
    check(lib.gnm_agg_fwd_bnrelu(ar.rowptr.buf.data_ptr(), ar.col.buf.data_ptr(), batch.rp_off.data_ptr(),
                                 batch.col_off.data_ptr(), batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max,
                                 zd.data_ptr(), F, scd.data_ptr(), shd.data_ptr(), hout.data_ptr(), F,
                                 gwin.data_ptr(), gf.stride(0), graph_avg, y.data_ptr(), F, F,
                                 epsd.data_ptr() if learn_eps else None, average, int(not learn_eps), _stream()),
          "gnm_agg_fwd_bnrelu")
    h32 = np.maximum(z * sc + sh, np.float32(0))                    # fp32, as the kernel forms it
    assert_close(hout.cpu().numpy(), h32.astype(np.float64), rtol=1e-6, what="h = relu(bn(z))")
    h = hout.cpu().numpy().astype(np.float64)
    off = np.concatenate([[0], np.cumsum(sizes)])
    ref_g = np.stack([h[off[g]:off[g + 1]].sum(0) / (sizes[g] if graph_avg else 1) for g in range(B)])
    assert_close(gwin.cpu().numpy(), ref_g, rtol=TOL, what="readout")
    assert torch.isnan(gf[:, :F]).all() and torch.isnan(gf[:, 2 * F:]).all()        # nothing outside the window
    deg = np.asarray(A.sum(1)).reshape(-1, 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        if learn_eps:
            ref = A @ h
            if average:
                ref = ref / deg
            ref = ref + (1 + eps) * h
        else:
            ref = A @ h + h
            if average:
                ref = ref / (deg + 1)
    assert_close(y.cpu().numpy(), ref, rtol=TOL, what="aggregation of the fused activation")


def test_agg_fused_bn_relu_refuses_other_shapes():
    from gnm._cabi import lib
    from gnm.arena import GraphArena
    rng = np.random.default_rng(0)
    ar = GraphArena(DEV)
    batch = ar.batch(random_graphs(rng, [30, 30], 0.3, True))
    z = torch.zeros((60, 48), device=DEV)            # (neither the 64-wide shape nor whole 32-float slices)
    v = torch.zeros(48, device=DEV)
    assert lib.gnm_agg_fwd_bnrelu(ar.rowptr.buf.data_ptr(), ar.col.buf.data_ptr(), batch.rp_off.data_ptr(),
                                  batch.col_off.data_ptr(), batch.node_off.data_ptr(), 2, batch.n_max, batch.nnz_max,
                                  z.data_ptr(), 48, v.data_ptr(), v.data_ptr(), z.data_ptr(), 48, None, 0, 0,
                                  z.data_ptr(), 48, 48, None, 0, 1, _stream()) == -2
    # a 32-float slice that fills LDS to the last kilobyte (1231 nodes) leaves no room for the readout shares: declined
    # before anything is launched (the caller then runs gnm_bn_relu_readout + gnm_agg)
    ar2 = GraphArena(DEV)
    b2 = ar2.batch(random_graphs(rng, [1231, 5], 0.01, True))
    assert lib.gnm_agg_slice_width(128, b2.n_max) == 32
    z2 = torch.zeros((b2.N, 128), device=DEV)
    v2 = torch.zeros(128, device=DEV)
    y2 = torch.full((b2.N, 128), float("nan"), device=DEV)
    assert lib.gnm_agg_fwd_bnrelu(ar2.rowptr.buf.data_ptr(), ar2.col.buf.data_ptr(), b2.rp_off.data_ptr(),
                                  b2.col_off.data_ptr(), b2.node_off.data_ptr(), 2, b2.n_max, b2.nnz_max,
                                  z2.data_ptr(), 128, v2.data_ptr(), v2.data_ptr(), None, 0, None, 0, 0,
                                  y2.data_ptr(), 128, 128, None, 0, 1, _stream()) == -2
    assert torch.isnan(y2).all()


@pytest.mark.parametrize("N,F,pad", [(1200, 7, 0), (1203, 7, 0), (37, 64, 0), (500, 7, 5), (1, 3, 0)])
def test_rowdot_partials(N, F, pad):
    """gnm_rowdot_partials (d eps of a layer whose aggregation backward has no other consumer): sum_v A[v,:].B[v,:]
    in fp64 partials, dense (flat 16-B path) and strided operands."""
    from gnm._cabi import check, lib
    rng = np.random.default_rng(N + F)
    A = torch.from_numpy(rng.standard_normal((N, F + pad)).astype(np.float32)).to(DEV)
    Bm = torch.from_numpy(rng.standard_normal((N, F + pad)).astype(np.float32)).to(DEV)
    part = torch.full((lib.gnm_rowdot_num_partials(),), float("nan"), dtype=torch.float64, device=DEV)
    check(lib.gnm_rowdot_partials(A.data_ptr(), F + pad, Bm.data_ptr(), F + pad, N, F, part.data_ptr(), _stream()), "rowdot")
    a, b = A[:, :F].cpu().numpy().astype(np.float64), Bm[:, :F].cpu().numpy().astype(np.float64)
    want, scale = float((a * b).sum()), float(np.abs(a * b).sum())
    assert abs(float(part.sum().item()) - want) <= 2e-6 * scale


# ------------------------------------------------------------------ graph-level head
@pytest.mark.parametrize("B,L,H,C,p,with_T", [(3, 5, 64, 2, 0.5, True), (4, 3, 32, 2, 0.0, True), (1, 2, 32, 3, 0.3, False),
                                              (257, 5, 128, 2, 0.5, True), (5, 1, 20, 7, 0.0, False),
                                              (1024, 5, 64, 2, 0.5, True), (9, 2, 200, 3, 0.0, True)])
def test_head_fwd_bwd_vs_numpy(B, L, H, C, p, with_T):
    """gnm_head_fwd / gnm_head_bwd against the per-layer restatement of graphcnn.py:224-231,239 in fp64."""
    import ctypes as CT
    from gnm._cabi import check, lib, ptr
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    rng = np.random.default_rng(B * 100 + L * 10 + C)
    LH = L * H
    g_f = rng.standard_normal((B, LH)).astype(np.float32) * 3
    Wp = [rng.standard_normal((C, H)).astype(np.float32) for _ in range(L)]
    bp = [rng.standard_normal(C).astype(np.float32) for _ in range(L)]
    masks = None
    if p > 0:
        masks = ((rng.random((L, B, C)) >= p) / (1 - p)).astype(np.float32)
    dC = rng.standard_normal((B, C)).astype(np.float32)
    T = rng.standard_normal((B, LH)).astype(np.float32) if with_T else None
    tg, tdC = t(g_f), t(dC)
    tW, tb = [t(w) for w in Wp], [t(b) for b in bp]
    tm = t(masks) if masks is not None else None
    tT = t(T) if T is not None else None
    wp = (CT.c_void_p * L)(*[w.data_ptr() for w in tW])
    bpp = (CT.c_void_p * L)(*[b.data_ptr() for b in tb])
    c_logit = torch.empty((B, C), device=DEV)
    csig = torch.empty((B, LH), device=DEV)
    check(lib.gnm_head_fwd(tg.data_ptr(), LH, B, L, H, C, wp, bpp, ptr(tm), c_logit.data_ptr(), C, csig.data_ptr(), LH,
                           _stream()), "head fwd")
    M = masks.astype(np.float64) if masks is not None else np.ones((L, B, C))
    g64 = g_f.astype(np.float64)
    ref_c = sum(M[l] * (g64[:, l * H:(l + 1) * H] @ Wp[l].astype(np.float64).T + bp[l]) for l in range(L))
    ref_sig = 1 / (1 + np.exp(-g64))
    assert_close(c_logit.cpu().numpy(), ref_c, rtol=TOL, what="c_logit")
    assert_close(csig.cpu().numpy(), ref_sig, rtol=TOL, what="sigmoid(g_f)")
    dW = [torch.full((C, H), 7.0, device=DEV) for _ in range(L)]
    db = [torch.full((C,), 7.0, device=DEV) for _ in range(L)]
    dwp = (CT.c_void_p * L)(*[w.data_ptr() for w in dW])
    dbpp = (CT.c_void_p * L)(*[b.data_ptr() for b in db])
    dph = torch.empty((B, LH), device=DEV)
    check(lib.gnm_head_bwd(tdC.data_ptr(), C, ptr(tm), tg.data_ptr(), LH, csig.data_ptr(), LH, ptr(tT), LH, B, L, H, C,
                           wp, dwp, dbpp, dph.data_ptr(), LH, _stream()), "head bwd")
    ref_dph = np.zeros((B, LH))
    for l in range(L):
        dlg = dC.astype(np.float64) * M[l]
        ref_dph[:, l * H:(l + 1) * H] = dlg @ Wp[l].astype(np.float64)
        assert_close(dW[l].cpu().numpy(), dlg.T @ g64[:, l * H:(l + 1) * H], rtol=TOL, what=f"dWp[{l}]")
        assert_close(db[l].cpu().numpy(), dlg.sum(0), rtol=TOL, what=f"dbp[{l}]", floor=1e-3)
    if T is not None:
        ref_dph += T.astype(np.float64) * ref_sig * (1 - ref_sig)
    assert_close(dph.cpu().numpy(), ref_dph, rtol=TOL, what="dph")


def test_head_unsupported_shape_is_reported():
    import ctypes as CT
    from gnm._cabi import lib
    L, H, Cn = 2, 8, 300                      # more classes than the head kernel's 256 threads
    g = torch.zeros((2, L * H), device=DEV)
    w = [torch.zeros((Cn, H), device=DEV) for _ in range(L)]
    b = [torch.zeros(Cn, device=DEV) for _ in range(L)]
    wp = (CT.c_void_p * L)(*[x.data_ptr() for x in w])
    bp = (CT.c_void_p * L)(*[x.data_ptr() for x in b])
    out = torch.zeros((2, Cn), device=DEV)
    assert lib.gnm_head_fwd(g.data_ptr(), L * H, 2, L, H, Cn, wp, bp, None, out.data_ptr(), Cn, None, 0, _stream()) == -2


# ------------------------------------------------------------------ full-size properties
@pytest.fixture(scope="module")
def full_batch():
    from gnm import synth
    from gnm.arena import GraphArena
    pool = synth.make_pool("dense_fc", 32)
    ar = GraphArena(DEV)
    gids = np.array([ar.add(g) for g in pool], dtype=np.int64)
    rng = np.random.default_rng(0)
    batch = ar.batch_from_gids(gids[rng.integers(0, 32, 1024)])
    return pool, ar, batch


@pytest.mark.parametrize("path", ["matrix-core", "gather"])
def test_full_size_aggregation_properties(full_batch, path):
    """B = 1024 x (n = 400, E = 47,600): exact edge count, linearity, the degree-weighted
    column checksum sum_v (A x)_v = sum_u deg(u) x_u (symmetric graphs), and determinism -- on the kernel these dense
    graphs take by default (csrc/aggm.hip) and on the CSR gather (csrc/agg.hip)."""
    import copy
    from gnm import core
    pool, ar, batch = full_batch
    assert batch.dense
    if path == "gather":
        batch = copy.copy(batch)
        batch.dense = False
    assert all(int(g.edge_mat.shape[1]) == 47600 for g in pool)
    N, F = batch.N, 64
    assert N == 409600
    spec = core.GinSpec(1, 1, True, "sum", "sum")
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N, F, device=DEV, generator=g)
    y = torch.randn(N, F, device=DEV, generator=g)
    zero_eps = torch.tensor([-1.0], device=DEV)            # (1 + eps) = 0: pure neighbour sum
    out = lambda v: (lambda o: (core._agg(batch, v, o, F, zero_eps.data_ptr(), spec, False), o)[1])(torch.empty_like(v))
    ax, ay, axy = out(x), out(y), out(2.5 * x - 0.5 * y)
    lin = 2.5 * ax - 0.5 * ay
    assert (axy - lin).abs().max().item() <= 1e-5 * lin.abs().max().item()
    assert torch.equal(out(x), ax)                        # fixed summation order: bitwise reproducible
    # checksum of checksums in fp64 on the device
    rp = ar.rowptr.buf[: ar.rowptr.size].to(torch.int64)
    deg = torch.cat([(rp[o + 1:o + 401] - rp[o:o + 400]) for o in batch.rp_off.tolist()[:64]])   # first 64 graphs
    lhs = ax[: 64 * 400].double().sum(0)
    rhs = (deg.double()[:, None] * x[: 64 * 400].double()).sum(0)
    assert (lhs - rhs).abs().max().item() <= 1e-6 * rhs.abs().max().item() + 1e-3


def test_full_size_model_step_runs_and_is_deterministic(full_batch):
    """One full training step at BASELINE configs[1] size; six runs agree bitwise.  (Only this size runs the
    persistent-grid loops of the Linear kernels for several tiles per wave and keeps all 256 CUs busy: it is the test
    that caught the buffer-store hazard of the streaming kernel -- wrong values in ~1 % of the tiles, different ones in
    every launch -- which no small case could show.)"""
    from models.graphcnn import GIN_InfoMaxReg
    pool, ar, _ = full_batch
    outs = []
    for _ in range(6):
        torch.manual_seed(0)
        model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.0, True, "sum", "sum", torch.device(DEV)).to(DEV).train()
        arena = model.arena()
        gids = np.array([arena.add(g) for g in pool], dtype=np.int64)
        batch = arena.batch_from_gids(gids[np.random.default_rng(0).integers(0, 32, 1024)])
        c, d = model.forward_batch(batch, perm=np.random.default_rng(1).permutation(1024))
        assert c.shape == (1024, 2) and d.shape == (2 * 409600, 1)
        (c.square().mean() + d.square().mean()).backward()
        outs.append([c.detach().clone(), d.detach().clone()] + [p.grad.clone() for p in model.parameters()])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("N", [150001, 98304])
def test_linear_kernels_many_tiles_per_wave(N):
    """The persistent-grid Linear kernels at a size where every wave walks several 32-row tiles (the small cases above
    give each wave at most one, i.e. only the peeled first iteration of gnm_lin_stream_kernel and
    gnm_linear_bwd_pipe_kernel): forward with prologue + statistics, fused backward without statistics (pipelined
    kernel) and the second-Linear form (lower BatchNorm's mask and sums from the operand registers), against fp64 torch
    products on the GPU.  150,001 rows end in a partial tile.  dW / db sum 1.5e5 fp32 products: 5e-5."""
    from gnm import core
    from gnm._cabi import check, lib
    K = H = 64
    g = torch.Generator(device=DEV).manual_seed(N)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    X, W, b = rn(N, K), rn(H, K) / 8, rn(H)
    sc, sh = torch.rand(K, device=DEV, generator=g) + 0.5, rn(K) * 0.3
    Z = torch.full((N, H), float("nan"), device=DEV)
    grid = lib.gnm_linear_grid(N)
    stats = torch.zeros((grid, 2, H), dtype=torch.float64, device=DEV)
    core._linear(X, W, 0, b, Z, N, K, H, (sc, sh), stats)
    Xe = torch.relu(X * sc + sh).double()
    ref = Xe @ W.double().t() + b.double()
    assert_close(Z.cpu().numpy(), ref.cpu().numpy(), rtol=TOL, what="linear fwd")
    st = stats.sum(0)
    assert_close(st[0].cpu().numpy(), Z.double().sum(0).cpu().numpy(), rtol=1e-6, what="column sums")
    assert_close(st[1].cpu().numpy(), (Z.double() ** 2).sum(0).cpu().numpy(), rtol=1e-6, what="column sums of squares")
    # fused backward
    G, Zb = rn(N, H), rn(N, H) * 1.5 + 0.5
    mean, rstd = Zb.mean(0), 1 / torch.sqrt(Zb.var(0, unbiased=False) + 1e-5)
    cA, m1, m2 = torch.rand(H, device=DEV, generator=g) + 0.5, G.mean(0), rn(H) * 0.1
    dZ = cA.double() * (G.double() - m1.double() - (Zb.double() - mean.double()) * rstd.double() * m2.double())
    ws = torch.empty(int(lib.gnm_linear_bwd_workspace_floats(N, H, K)), device=DEV)
    lmean, lrstd = rn(K) * 0.1, torch.rand(K, device=DEV, generator=g) + 0.5
    for form in ("no statistics, raw input", "second Linear of an MLP"):
        second = form.startswith("second")
        dA = torch.full((N, K), float("nan"), device=DEV)
        dW = torch.full((H, K), float("nan"), device=DEV)
        db = torch.full((H,), float("nan"), device=DEV)
        lp = torch.full((lib.gnm_linear_bwd_grid(N), 2, K), float("nan"), dtype=torch.float64, device=DEV)
        check(lib.gnm_linear_bwd_fused(
            G.data_ptr(), H, Zb.data_ptr(), H, mean.data_ptr(), rstd.data_ptr(), cA.data_ptr(), m1.data_ptr(),
            m2.data_ptr(), X.data_ptr(), K, sc.data_ptr() if second else None, sh.data_ptr() if second else None,
            1 if second else 0, W.data_ptr(), K, dA.data_ptr(), K, dW.data_ptr(), K, db.data_ptr(), ws.data_ptr(), N, K, H,
            X.data_ptr() if second else None, K if second else 0, sc.data_ptr() if second else None,
            sh.data_ptr() if second else None, lmean.data_ptr() if second else None,
            lrstd.data_ptr() if second else None, lp.data_ptr() if second else None, _stream()), form)
        Xin = Xe if second else X.double()
        dX_ref = dZ @ W.double()
        if second:
            mask = (X * sc + sh > 0)
            dX_ref = dX_ref * mask
            s_ref = torch.stack([dX_ref.sum(0), (dX_ref * ((X.double() - lmean.double()) * lrstd.double())).sum(0)])
            fl = 1e-3 * float(dX_ref.abs().sum(0).max())
            assert_close(lp.sum(0).cpu().numpy(), s_ref.cpu().numpy(), rtol=TOL, what=form + ": lower BatchNorm sums", floor=fl)
        assert_close(dA.cpu().numpy(), dX_ref.cpu().numpy(), rtol=TOL, what=form + ": dX")
        assert_close(dW.cpu().numpy(), (dZ.t() @ Xin).cpu().numpy(), rtol=5e-5, what=form + ": dW")
        assert_close(db.cpu().numpy(), dZ.sum(0).cpu().numpy(), rtol=5e-5, what=form + ": db",
                     floor=1e-2 * float(dZ.abs().sum(0).max()))


@pytest.mark.parametrize("average,learn_eps", [(0, 1), (1, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize("F", [16, 64, 7])
def test_aggregation_of_graphs_too_large_for_lds(average, learn_eps, F):
    """Graphs beyond ~4,500 nodes have no LDS-resident slice: gnm_agg gathers their neighbour rows from global memory
    (gnm_agg_global_kernel; SURVEY.md 8(d), sparse row) instead of refusing them.  Forward, backward (transposed
    structure of an asymmetric graph) and the d-eps partials against the fp64 restatement of graphcnn.py:154-161 /
    178-182; mixed with a small graph in the same batch (n_max decides the kernel for the whole launch)."""
    from gnm import core
    from gnm._cabi import lib
    from gnm.arena import GraphArena
    rng = np.random.default_rng(F * 10 + average)
    n_big = 6000
    assert lib.gnm_agg_slice_width(F, n_big) == 0
    src = rng.integers(0, n_big, 30000)
    dst = rng.integers(0, n_big, 30000)
    keep = src != dst
    em = np.unique(np.stack([src[keep], dst[keep]]), axis=1)            # asymmetric, no repeated edges
    graphs = [RG(n_big, em, 3, rng)] + random_graphs(rng, [50], 0.2, False)
    ar = GraphArena(DEV)
    batch = ar.batch(graphs)
    assert not batch.dense and not batch.symmetric
    A = dense_adj(graphs)
    N = batch.N
    deg = np.asarray(A.sum(1)).reshape(-1, 1)
    x = rng.standard_normal((N, F)).astype(np.float32)
    eps = 0.37
    spec = core.GinSpec(1, 1, bool(learn_eps), "sum", "average" if average else "sum")
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.full((N, F), float("nan"), device=DEV)
    epsd = torch.tensor([eps], device=DEV)
    core._agg(batch, xd, yd, F, epsd.data_ptr() if learn_eps else None, spec, backward=False)
    x64 = x.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        if learn_eps:
            ref = A @ x64
            if average:
                ref = ref / deg
            ref = ref + (1 + np.float32(eps).astype(np.float64)) * x64
        else:
            ref = (A @ x64 + x64) / ((deg + 1) if average else 1.0)
    assert_close(yd.cpu().numpy(), ref, rtol=TOL, what="global gather forward")
    # backward: d h = A^T (d pooled [/ deg]) + self term; d eps partial = sum d pooled . h
    dp = rng.standard_normal((N, F)).astype(np.float32)
    dpd = torch.from_numpy(dp).to(DEV)
    dhd = torch.full((N, F), float("nan"), device=DEV)
    part = torch.full((core.agg_partials_capacity(batch, F),), float("nan"), dtype=torch.float64, device=DEV)
    cnt = core._agg(batch, dpd, dhd, F, epsd.data_ptr() if learn_eps else None, spec, backward=True,
                    hfwd=xd if learn_eps else None, deps_partial=part if learn_eps else None)
    dp64 = dp.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        if learn_eps:
            g = dp64 / deg if average else dp64
            g = np.where(deg > 0, g, 0.0) if average else g
            ref_b = A.T @ g + (1 + np.float32(eps).astype(np.float64)) * dp64
        else:
            g = dp64 / (deg + 1) if average else dp64
            ref_b = A.T @ g + g
    assert_close(dhd.cpu().numpy(), ref_b, rtol=TOL, what="global gather backward")
    if learn_eps:
        assert cnt == lib.gnm_agg_num_partials(F, batch.n_max, batch.B) > 0
        want = float((dp64 * x64).sum())
        assert abs(float(part[:cnt].sum().item()) - want) <= 1e-6 * float(np.abs(dp64 * x64).sum())



@pytest.mark.parametrize("M,N,K", [(1024, 320, 320), (37, 64, 96), (1, 32, 32), (320, 320, 1000), (33, 40, 20), (64, 640, 640)])
def test_small_gemm_all_operand_forms(M, N, K):
    """gnm_small_gemm (csrc/sgemm.hip), the tail's three products: every operand form (row-major / column-major A and
    B), sizes that are not multiples of the 32 x 32 tile or of the 16-wide step, an operand that is not 16-byte aligned
    (a parameter inside the flat buffer), element by element below 1e-6 x sum_k |a||b| against fp64, and bitwise
    reproducible."""
    from gnm._cabi import check, lib
    rng = np.random.default_rng(M + 7 * N + 13 * K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Bm = rng.standard_normal((K, N)).astype(np.float32)
    ref = A.astype(np.float64) @ Bm.astype(np.float64)
    bound = np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64) + 1e-30
    for a_cols in (0, 1):
        for b_cols in (0, 1):
            for off in (0, 3):
                flat_a = torch.zeros(off + M * K, device=DEV)
                Ad = flat_a[off:].view((K, M) if a_cols else (M, K))
                Ad.copy_(torch.from_numpy(np.ascontiguousarray(A.T if a_cols else A)))
                flat_b = torch.zeros(off + K * N, device=DEV)
                Bd = flat_b[off:].view((K, N) if b_cols else (N, K))
                Bd.copy_(torch.from_numpy(np.ascontiguousarray(Bm if b_cols else Bm.T)))
                outs = []
                for _ in range(2):
                    C_ = torch.full((M, N + 5), float("nan"), device=DEV)
                    check(lib.gnm_small_gemm(Ad.data_ptr(), Ad.stride(0), a_cols, Bd.data_ptr(), Bd.stride(0), b_cols,
                                             C_.data_ptr(), N + 5, M, N, K, _stream()), "gnm_small_gemm")
                    outs.append(C_.cpu().numpy())
                assert np.isnan(outs[0][:, N:]).all()
                err = np.abs(outs[0][:, :N] - ref) / bound
                assert err.max() < 1e-6, (a_cols, b_cols, off, err.max())
                assert np.array_equal(outs[0][:, :N], outs[1][:, :N])


@pytest.mark.parametrize("N", [1, 77, 5000 + 3])
def test_masked_dgrad_k128(N):
    """gnm_linear_dgrad_masked (K = H = 128): dX = dZ W with the lower ReLU's mask and the lower BatchNorm's backward sums
    in the epilogue, against fp64"""
    from gnm._cabi import check, lib
    K = H = 128
    rng = np.random.default_rng(N)
    dZ = rng.standard_normal((N, H)).astype(np.float32)
    W = (rng.standard_normal((H, K)) / 11).astype(np.float32)
    mZ = rng.standard_normal((N, K)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, K).astype(np.float32), (rng.standard_normal(K) * 0.3).astype(np.float32)
    mu, rs = rng.standard_normal(K).astype(np.float32), rng.uniform(0.5, 1.5, K).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    dZd, Wd, mZd, scd, shd, mud, rsd = map(t, (dZ, W, mZ, sc, sh, mu, rs))
    G = torch.full((N, K), float("nan"), device=DEV)
    grid = int(lib.gnm_linear_grid(N))
    part = torch.full((grid, 2, K), float("nan"), dtype=torch.float64, device=DEV)
    check(lib.gnm_linear_dgrad_masked(dZd.data_ptr(), H, Wd.data_ptr(), K, G.data_ptr(), K, N, K, H, mZd.data_ptr(), K,
                                      scd.data_ptr(), shd.data_ptr(), mud.data_ptr(), rsd.data_ptr(), part.data_ptr(),
                                      _stream()), "gnm_linear_dgrad_masked")
    ref = dZ.astype(np.float64) @ W.astype(np.float64)
    ref = ref * ((mZ * sc + sh) > 0)
    assert_close(G.cpu().numpy(), ref, rtol=TOL, what="masked dX")
    ps = part.sum(0).cpu().numpy()
    xhat = (mZ.astype(np.float64) - mu) * rs
    assert_close(ps[0], ref.sum(0), rtol=TOL, what="sum G", floor=np.abs(ref).sum(0).max() * 1e-6)
    assert_close(ps[1], (ref * xhat).sum(0), rtol=TOL, what="sum G xhat", floor=np.abs(ref * xhat).sum(0).max() * 1e-6)


@pytest.mark.parametrize("N,K,want_dx", [(1, 7, True), (100, 7, True), (4096 + 33, 7, True), (40000 + 5, 3, True), (5000, 16, False)])
def test_narrow_linear_backward_recomputing_its_output(N, K, want_dx):
    """gnm_linear_bwd_fused_rz for a narrow input (K <= 16, the input layer's first Linear): against fp64 and against
    gnm_linear_bwd_fused fed the Z that gnm_linear_fwd wrote; wider narrow inputs are declined"""
    from gnm import core
    from gnm._cabi import check, lib
    H = 64
    rng = np.random.default_rng(N + K)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(DEV)
    X = (rng.standard_normal((N, K)) * 1.5 + 0.3).astype(np.float32)
    W = (rng.standard_normal((H, K)) / 3).astype(np.float32)
    b = rng.standard_normal(H).astype(np.float32)
    G = rng.standard_normal((N, H)).astype(np.float32)
    mean, rstd, cA, m1, m2 = (rng.uniform(0.5, 1.5, H).astype(np.float32) for _ in range(5))
    Xd, Wd, bd, Gd = map(t, (X, W, b, G))
    vec = [t(v) for v in (mean, rstd, cA, m1, m2)]
    Z = torch.empty(N, H, device=DEV)
    core._linear(Xd, Wd, 0, bd, Z, N, K, H, None, None)
    assert lib.gnm_linear_bwd_fused_rz(Gd.data_ptr(), H, bd.data_ptr(), *[v.data_ptr() for v in vec], Xd.data_ptr(), 31, None, None,
                                       0, Wd.data_ptr(), 31, None, 0, None, 0, None, None, N, 31, H, None, 0, None, None, None,
                                       None, None, _stream()) == -2          # K = 31: two steps of Z, not this kernel's
    out = []
    for rz in (True, False):
        dA = torch.full((N, K), float("nan"), device=DEV) if want_dx else None
        dW = torch.full((H, K), float("nan"), device=DEV)
        db = torch.full((H,), float("nan"), device=DEV)
        ws = torch.empty(int(lib.gnm_linear_bwd_workspace_floats(N, H, K)), device=DEV)
        tail = (Xd.data_ptr(), K, None, None, 0, Wd.data_ptr(), K, dA.data_ptr() if want_dx else None, K if want_dx else 0,
                dW.data_ptr(), K, db.data_ptr(), ws.data_ptr(), N, K, H, None, 0, None, None, None, None, None, _stream())
        if rz:
            check(lib.gnm_linear_bwd_fused_rz(Gd.data_ptr(), H, bd.data_ptr(), *[v.data_ptr() for v in vec], *tail), "rz")
        else:
            check(lib.gnm_linear_bwd_fused(Gd.data_ptr(), H, Z.data_ptr(), H, *[v.data_ptr() for v in vec], *tail), "fused")
        out.append([a.cpu().numpy() if a is not None else None for a in (dA, dW, db)])
    X64 = X.astype(np.float64)
    Z64 = X64 @ W.astype(np.float64).T + b
    dZ = cA * (G - m1 - (Z64 - mean) * rstd * m2)
    for name, (dA, dW, db) in zip(("rz", "stored Z"), out):
        if want_dx:
            assert_close(dA, dZ @ W.astype(np.float64), rtol=TOL, what=name + " dX")
        assert_close(dW, dZ.T @ X64, rtol=TOL, what=name + " dW")
        assert_close(db, dZ.sum(0), rtol=TOL, what=name + " db")


@pytest.mark.parametrize("sizes,width,pad", [([5, 1, 12, 400], 7, 0), ([3] * 40, 7, 1), ([1000, 17], 128, 0), ([9], 3, 5)])
@pytest.mark.parametrize("second", [False, True])
def test_gather_graph_rows(sizes, width, pad, second):
    """gnm_gather_graph_rows (X_concat, graphcnn.py:195): every graph's rows copied from its block of the store to its
    place in the batch, a second array alongside; nothing else of the destination is touched."""
    from gnm._cabi import check, lib
    rng = np.random.default_rng(len(sizes) + width)
    B, N = len(sizes), int(sum(sizes))
    lds, ldd = width + pad, width + 2 * pad
    # the store keeps the graphs in another order, with gaps between their blocks
    order = rng.permutation(B)
    base = np.zeros(B, dtype=np.int64)
    at = 3
    for g in order:
        base[g] = at
        at += sizes[g] + int(rng.integers(0, 4))
    store = rng.standard_normal((at + 2, lds)).astype(np.float32)
    store2 = rng.standard_normal((at + 2, lds)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    sd, s2d, based = t(store), t(store2), t(base)
    node_off = t(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32))
    dst = torch.full((N, ldd), float("nan"), device=DEV)
    dst2 = torch.full((N, ldd), float("nan"), device=DEV)
    check(lib.gnm_gather_graph_rows(sd.data_ptr(), s2d.data_ptr() if second else None, lds, width, based.data_ptr(),
                                    node_off.data_ptr(), B, dst.data_ptr(), dst2.data_ptr() if second else None, ldd,
                                    _stream()), "gnm_gather_graph_rows")
    rows = np.concatenate([np.arange(base[g], base[g] + sizes[g]) for g in range(B)])
    assert np.array_equal(dst[:, :width].cpu().numpy(), store[rows][:, :width])
    assert torch.isnan(dst[:, width:]).all()
    if second:
        assert np.array_equal(dst2[:, :width].cpu().numpy(), store2[rows][:, :width])
    else:
        assert torch.isnan(dst2).all()
    assert lib.gnm_gather_graph_rows(sd.data_ptr(), s2d.data_ptr(), lds, width, based.data_ptr(), node_off.data_ptr(), B,
                                     dst.data_ptr(), None, ldd, _stream()) == -1          # a second source needs its destination
