"""The matrix-core aggregation (csrc/aggm.hip: bit adjacency x three bf16 planes of the features) through the C-ABI:
against the fp64 restatement of graphcnn.py:154-161 / 178-182 at the library's fp32 tolerance (1e-5 relative, max
norm), against the CSR gather kernels it stands in for (same fused forms, same outputs), and its own invariants
(bit matrix == adjacency, refusals, run-to-run determinism)."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_close
from test_gpu_kernels import RG, dense_adj, random_graphs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_bit_adjacency_matches_the_csr():
    from gnm._cabi import lib
    from gnm.arena import GraphArena
    rng = np.random.default_rng(5)
    graphs = random_graphs(rng, [37, 400, 1, 64, 416], 0.3, True) + random_graphs(rng, [45, 33], 0.3, False)
    # a multigraph: edge (0, 1) listed twice
    em = np.array([[0, 1, 0, 1], [1, 0, 1, 0]])
    graphs.append(RG(5, em, 3, rng))
    # too large for the kernel
    graphs += random_graphs(rng, [417], 0.05, True)
    ar = GraphArena(DEV)
    gids = ar.add_many(graphs)
    bits = ar.bits.buf[: ar.bits.size].cpu().numpy().view(np.uint32)
    for k, g in enumerate(graphs):
        gid = gids[k]
        n = len(g.g)
        if n > lib.gnm_aggm_max_nodes():
            assert not ar.bits_ok[gid]
            continue
        W = (n + 31) // 32
        HP = ((W + 1) // 2 + 3) // 4 * 4                                   # words per half row (include/gnm_hip.h)
        assert lib.gnm_adj_bits_words(n) == W * 32 * 2 * HP
        A = np.zeros((n, n), dtype=bool)
        e = g.edge_mat.numpy()
        A[e[0], e[1]] = True
        for off, M in ((ar.bits_off[gid], A), (ar.t_bits_off[gid], A.T)):
            w = bits[off:off + W * 32 * 2 * HP].reshape(32 * W, 2 * HP)
            by = w.view(np.uint8).reshape(32 * W, 2, 4 * HP)                # [row][half][position]
            nb = 4 * W                                                      # bytes of a row: byte j -> half j & 1, position j >> 1
            rowbytes = np.zeros((32 * W, nb), dtype=np.uint8)
            rowbytes[:, 0::2] = by[:, 0, :(nb + 1) // 2]
            rowbytes[:, 1::2] = by[:, 1, :nb // 2]
            assert not by[:, 0, (nb + 1) // 2:].any() and not by[:, 1, nb // 2:].any()     # padding stays zero
            got = np.unpackbits(rowbytes, axis=1, bitorder="little").astype(bool)           # [32 W, 32 W]
            assert np.array_equal(got[:n, :n], M)
            assert not got[n:].any() and not got[:, n:].any()             # padding rows / bits stay zero
        assert ar.bits_ok[gid] == (k != len(graphs) - 2)                   # only the multigraph is refused
        assert (ar.t_bits_off[gid] == ar.bits_off[gid]) == ar.sym[gid]
    # a batch with the multigraph in it stays on the gather
    assert not ar.batch_from_gids(np.array(gids[5:8], dtype=np.int64)).dense
    assert ar.batch_from_gids(np.array(gids[:2], dtype=np.int64)).dense


AGGM_CASES = [
    # (sizes, density, F, symmetric)
    ([400, 400, 400], 0.3, 64, True),        # headline shape
    ([40, 40, 40], 0.3, 64, True),
    ([37, 5, 64, 1, 23, 416, 399, 33], 0.4, 64, True),   # ragged, single node, the largest supported graph
    ([200, 200], 1.0, 64, True),             # complete graphs
    ([64, 64], 0.2, 32, True),               # one column block
    ([300, 130], 0.15, 128, True),           # four column blocks
    ([45, 45, 45], 0.3, 64, False),          # asymmetric: backward runs on the transposed bit matrix
    ([17] * 11, 0.5, 64, True),              # more graphs than one XCD round
    ([17] * 300 + [33, 1, 64], 0.5, 64, True),   # 606 workgroups: several rounds of the chip
    ([400, 400], 0.3, 7, True),              # the input layer: one partial column block, unaligned rows
    ([37, 5, 64, 1, 23], 0.4, 5, True),
    ([45, 45, 45], 0.3, 20, False),
]


@pytest.mark.parametrize("sizes,density,F,symmetric", AGGM_CASES)
@pytest.mark.parametrize("average,learn_eps", [(0, 1), (1, 1), (0, 0), (1, 0)])
def test_aggm_forward_backward(sizes, density, F, symmetric, average, learn_eps):
    from gnm import core
    from gnm.arena import GraphArena
    rng = np.random.default_rng(hash((tuple(sizes), F, 7)) % 2**31)
    graphs = random_graphs(rng, sizes, density, symmetric)
    ar = GraphArena(DEV)
    batch = ar.batch(graphs)
    assert batch.dense and core._dense(batch, F)
    # this test drives the matrix-core kernel itself, isolated nodes included (their 0/0 under average + learn_eps
    # stays in the row: the inputs are finite).  The model-level routing of such batches to the CSR gather
    # (core._dense with a spec) is switched off here and tested in test_dense_routing_of_isolated_nodes below.
    has_iso = batch.iso
    batch.iso = False
    A = dense_adj(graphs)
    N = batch.N
    deg = np.asarray(A.sum(1)).reshape(-1, 1)
    assert has_iso == bool((deg == 0).any())
    x = rng.standard_normal((N, F)) * np.exp(rng.standard_normal((N, 1)) * 2)     # rows of very different scale
    eps = 0.37
    spec = core.GinSpec(1, 1, bool(learn_eps), "sum", "average" if average else "sum")
    xd = torch.from_numpy(x.astype(np.float32)).to(DEV)
    yd = torch.full((N, F), float("nan"), device=DEV)
    epsd = torch.tensor([eps], device=DEV)
    core._agg(batch, xd, yd, F, epsd.data_ptr() if learn_eps else None, spec, backward=False)
    x32 = xd.cpu().numpy().astype(np.float64)
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
    assert_close(yd.cpu().numpy(), ref, rtol=TOL, what="aggm forward")
    # the CSR kernel on the same input: same result to fp32 rounding
    batch.dense = False
    y2 = torch.full((N, F), float("nan"), device=DEV)
    core._agg(batch, xd, y2, F, epsd.data_ptr() if learn_eps else None, spec, backward=False)
    batch.dense = True
    assert_close(yd.cpu().numpy(), y2.cpu().numpy().astype(np.float64), rtol=TOL, what="aggm vs gather")
    # determinism
    y3 = torch.full((N, F), float("nan"), device=DEV)
    core._agg(batch, xd, y3, F, epsd.data_ptr() if learn_eps else None, spec, backward=False)
    assert torch.equal(torch.nan_to_num(y3), torch.nan_to_num(yd))
    # backward
    dp = rng.standard_normal((N, F)).astype(np.float32)
    dpd = torch.from_numpy(dp).to(DEV)
    dhd = torch.full((N, F), float("nan"), device=DEV)
    part = torch.full((core.agg_partials_capacity(batch, F),), float("nan"), dtype=torch.float64, device=DEV)
    cnt = core._agg(batch, dpd, dhd, F, epsd.data_ptr() if learn_eps else None, spec, backward=True,
                    hfwd=xd if learn_eps else None, deps_partial=part if learn_eps else None)
    if F % 32 == 0:
        assert cnt == core.lib.gnm_aggm_num_partials(F, batch.B) == batch.B * (F // 32)
    elif learn_eps:      # narrow F with d eps: the matrix-core form declines, the CSR gather ran
        assert cnt == core.lib.gnm_agg_num_partials(F, batch.n_max, batch.B)
    dp64 = dp.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        if learn_eps:
            src = dp64 / deg if average else dp64
            src = np.where(np.isfinite(src), src, 0.0)
            refb = A.T @ src + (1 + np.float32(eps).astype(np.float64)) * dp64
        else:
            src = dp64 / (deg + 1) if average else dp64
            refb = A.T @ src + src
    got = dhd.cpu().numpy()
    if average and learn_eps:
        # a row without neighbours divides by zero in the pre-scale (x / 0): inf or NaN rows are never gathered by
        # the CSR kernel, while a product with a zero bit would turn them into NaN everywhere -- the kernel must not
        iso = np.asarray(deg).reshape(-1) == 0
        assert np.isfinite(got[~iso]).all()
    assert_close(got, refb, rtol=TOL, what="aggm backward")
    if learn_eps:
        out = torch.empty(1, device=DEV)
        core.check(core.lib.gnm_sum_partials(part.data_ptr(), cnt, out.data_ptr(), _stream()), "sum")
        want = float((dp64 * x32).sum())
        scale = float(np.abs(dp64 * x32).sum())
        assert abs(out.item() - want) <= 1e-6 * scale


def _bwd_stats_args(rng, batch, F, disc):
    N, B = batch.N, batch.B
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    d = dict(dp=t(rng.standard_normal((N, F)).astype(np.float32)), hf=t(rng.standard_normal((N, F)).astype(np.float32)),
             Z=t((rng.standard_normal((N, F)) + 0.3).astype(np.float32)),
             sc=t(rng.uniform(0.5, 1.5, F).astype(np.float32)), sh=t((rng.standard_normal(F) * 0.3).astype(np.float32)),
             mu=t(rng.standard_normal(F).astype(np.float32) * 0.1), rs=t(rng.uniform(0.5, 1.5, F).astype(np.float32)),
             dpool=t(rng.standard_normal((B, F)).astype(np.float32)),
             dsc1=t(rng.standard_normal(N).astype(np.float32)) if disc else None,
             U=t(rng.standard_normal((B, 3 * F)).astype(np.float32)))
    perm = rng.permutation(B)
    inv = np.empty(B, dtype=np.int32); inv[perm] = np.arange(B, dtype=np.int32)
    d["inv"], d["s2"] = t(inv), t(rng.standard_normal(B).astype(np.float32))
    return d


@pytest.mark.parametrize("sizes,density", [([40, 40, 40], 0.3), ([400, 400], 0.3), ([64] * 9, 0.5),
                                           ([37, 5, 64, 1, 23, 416], 0.4), ([3] * 600, 0.9)])
@pytest.mark.parametrize("average,learn_eps,graph_avg,disc", [(0, 1, 0, True), (1, 1, 1, True), (0, 0, 0, False),
                                                              (1, 0, 1, True)])
def test_aggm_backward_fused_with_bn_stats(sizes, density, average, learn_eps, graph_avg, disc):
    """gnm_aggm_bwd_stats == gnm_agg_bwd_stats (the CSR form, itself held to gnm_agg + gnm_bn_relu_bwd_stats in
    test_gpu_kernels.py), to fp32 rounding; run-to-run bitwise.  [3] * 600: N < B rows take the shuffled-branch term
    in more than one graph."""
    from gnm._cabi import check, lib
    from gnm.arena import GraphArena
    rng = np.random.default_rng(len(sizes) * 1000 + average * 10 + learn_eps)
    graphs = random_graphs(rng, sizes, density, True)
    ar = GraphArena(DEV)
    batch = ar.batch(graphs)
    assert batch.dense
    N, B, F = batch.N, batch.B, 64
    d = _bwd_stats_args(rng, batch, F, disc)
    Ul = d["U"][:, F:2 * F]
    eps = torch.tensor([0.25], device=DEV)
    epsp = eps.data_ptr() if learn_eps else None
    a = ar

    def run(dense, hf_given, G, pf, part):
        tail = (d["Z"].data_ptr(), F, d["sc"].data_ptr(), d["sh"].data_ptr(), d["mu"].data_ptr(), d["rs"].data_ptr(),
                d["dpool"].data_ptr(), F, graph_avg, d["dsc1"].data_ptr() if disc else None,
                Ul.data_ptr() if disc else None, d["U"].stride(0) if disc else 0, d["inv"].data_ptr() if disc else None,
                d["s2"].data_ptr() if disc else None, pf.data_ptr(), _stream())
        mid = (d["dp"].data_ptr(), F, G.data_ptr(), F, F, epsp, average, int(not learn_eps),
               d["hf"].data_ptr() if hf_given else None, F if hf_given else 0, part.data_ptr() if part is not None else None)
        if dense:
            check(lib.gnm_aggm_bwd_stats(a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(),
                                         batch.t_col_off.data_ptr(), a.bits.buf.data_ptr(), batch.t_bits_off.data_ptr(),
                                         a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(), batch.node_off.data_ptr(), B,
                                         batch.n_max, *mid, *tail), "gnm_aggm_bwd_stats")
        else:
            check(lib.gnm_agg_bwd_stats(a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(),
                                        batch.t_col_off.data_ptr(), a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(),
                                        batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max, *mid, *tail),
                  "gnm_agg_bwd_stats")

    new = lambda: (torch.full((N, F), float("nan"), device=DEV),
                   torch.full((B, 2, F), float("nan"), dtype=torch.float64, device=DEV))
    # the unfused pair of kernels: gnm_agg (CSR) then gnm_bn_relu_bwd_stats
    from gnm import core
    spec = core.GinSpec(1, 1, bool(learn_eps), "average" if graph_avg else "sum", "average" if average else "sum")
    batch.dense = False
    dh = torch.empty((N, F), device=DEV)
    part_u = torch.empty(lib.gnm_agg_num_partials(F, batch.n_max, B), dtype=torch.float64, device=DEV)
    core._agg(batch, d["dp"], dh, F, epsp, spec, backward=True, hfwd=d["hf"] if learn_eps else None,
              deps_partial=part_u if learn_eps else None)
    batch.dense = True
    Gu, pu = new()
    check(lib.gnm_bn_relu_bwd_stats(dh.data_ptr(), F, d["dpool"].data_ptr(), F, graph_avg,
                                    d["dsc1"].data_ptr() if disc else None, Ul.data_ptr() if disc else None,
                                    d["U"].stride(0) if disc else 0, d["inv"].data_ptr() if disc else None,
                                    d["s2"].data_ptr() if disc else None, d["Z"].data_ptr(), F, d["sc"].data_ptr(),
                                    d["sh"].data_ptr(), d["mu"].data_ptr(), d["rs"].data_ptr(), 1, Gu.data_ptr(), F,
                                    batch.node_off.data_ptr(), B, F, pu.data_ptr(), _stream()), "stats")
    fl = 1e-3 * float(Gu.abs().sum(0).max())
    for hf_given in ([True, False] if learn_eps else [False]):
        want_part = bool(learn_eps)
        G0, p0 = new()
        part0 = torch.full((lib.gnm_agg_num_partials(F, batch.n_max, B),), float("nan"), dtype=torch.float64, device=DEV)
        run(False, hf_given, G0, p0, part0 if want_part else None)
        G1, p1 = new()
        part1 = torch.full((lib.gnm_aggm_num_partials(F, B),), float("nan"), dtype=torch.float64, device=DEV)
        run(True, hf_given, G1, p1, part1 if want_part else None)
        assert_close(G1.cpu().numpy(), Gu.cpu().numpy().astype(np.float64), rtol=TOL, what="G vs unfused")
        assert_close(p1.cpu().numpy(), pu.cpu().numpy(), rtol=TOL, what="BN-backward partial sums vs unfused", floor=fl)
        assert_close(G0.cpu().numpy(), Gu.cpu().numpy().astype(np.float64), rtol=TOL, what="CSR fused G vs unfused")
        assert_close(p0.cpu().numpy(), pu.cpu().numpy(), rtol=TOL, what="CSR fused partial sums vs unfused", floor=fl)
        if want_part:
            if hf_given:
                want = float(part_u.sum().item())
            else:
                h_re = torch.relu(d["Z"] * d["sc"] + d["sh"]).double()
                want = float((d["dp"].double() * h_re).sum().item())
            sc_ = float((d["dp"].double() * (d["hf"].double() if hf_given else h_re)).abs().sum().item()) + 1e-30
            assert abs(float(part1.sum().item()) - want) <= 2e-6 * sc_
            assert abs(float(part0.sum().item()) - want) <= 2e-6 * sc_
        G2, p2 = new()
        part2 = torch.full_like(part1, float("nan"))
        run(True, hf_given, G2, p2, part2 if want_part else None)
        assert torch.equal(G2, G1) and torch.equal(p2, p1) and (not want_part or torch.equal(part2, part1))


@pytest.mark.parametrize("sizes,density", [([40, 40, 40], 0.3), ([37, 5, 64, 1, 23, 416], 0.4), ([400, 400, 400], 0.3)])
@pytest.mark.parametrize("average,learn_eps,graph_avg", [(0, 1, 0), (1, 1, 1), (0, 0, 1), (1, 0, 0)])
def test_aggm_forward_with_fused_bn_relu_readout(sizes, density, average, learn_eps, graph_avg):
    from gnm._cabi import check, lib
    from gnm.arena import GraphArena
    rng = np.random.default_rng(len(sizes) * 1000 + sizes[0])
    graphs = random_graphs(rng, sizes, density, True)
    ar = GraphArena(DEV)
    batch = ar.batch(graphs)
    assert batch.dense
    A = dense_adj(graphs)
    N, F, B = batch.N, 64, batch.B
    z = rng.standard_normal((N, F)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, F).astype(np.float32)
    sh = (rng.standard_normal(F) * 0.3).astype(np.float32)
    eps = 0.37
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    zd, scd, shd, epsd = t(z), t(sc), t(sh), torch.tensor([eps], device=DEV)
    hout = torch.full((N, F), float("nan"), device=DEV)
    gf = torch.full((B, 3 * F), float("nan"), device=DEV)
    y = torch.full((N, F), float("nan"), device=DEV)
    gwin = gf[:, F:2 * F]
    check(lib.gnm_aggm_fwd_bnrelu(ar.rowptr.buf.data_ptr(), ar.col.buf.data_ptr(), batch.rp_off.data_ptr(),
                                  batch.col_off.data_ptr(), ar.bits.buf.data_ptr(), batch.bits_off.data_ptr(),
                                  batch.node_off.data_ptr(), B, batch.n_max, zd.data_ptr(), F, scd.data_ptr(),
                                  shd.data_ptr(), hout.data_ptr(), F, gwin.data_ptr(), gf.stride(0), graph_avg,
                                  y.data_ptr(), F, F, epsd.data_ptr() if learn_eps else None, average,
                                  int(not learn_eps), _stream()), "gnm_aggm_fwd_bnrelu")
    h32 = np.maximum(z * sc + sh, np.float32(0))
    assert_close(hout.cpu().numpy(), h32.astype(np.float64), rtol=1e-6, what="h = relu(bn(z))")
    h = hout.cpu().numpy().astype(np.float64)
    off = np.concatenate([[0], np.cumsum(sizes)])
    ref_g = np.stack([h[off[g]:off[g + 1]].sum(0) / (sizes[g] if graph_avg else 1) for g in range(B)])
    assert_close(gwin.cpu().numpy(), ref_g, rtol=TOL, what="readout")
    assert torch.isnan(gf[:, :F]).all() and torch.isnan(gf[:, 2 * F:]).all()
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


def test_aggm_refuses_what_it_does_not_cover():
    from gnm._cabi import lib
    from gnm.arena import GraphArena
    rng = np.random.default_rng(0)
    ar = GraphArena(DEV)
    batch = ar.batch(random_graphs(rng, [30, 30], 0.3, True))
    x = torch.zeros((60, 64), device=DEV)

    def call(n_max, F, ldx, bits=True):
        return lib.gnm_aggm(ar.rowptr.buf.data_ptr(), ar.col.buf.data_ptr(), batch.rp_off.data_ptr(),
                            batch.col_off.data_ptr(), ar.bits.buf.data_ptr() if bits else None, batch.bits_off.data_ptr(),
                            None, None, batch.node_off.data_ptr(), 2, n_max, x.data_ptr(), ldx, x.data_ptr(), ldx, F,
                            None, 0, 1, 0, None, 0, None, _stream())
    assert call(30, 64, 64) == 0
    assert call(417, 64, 64) == -2          # larger than the LDS planes allow
    assert call(30, 40, 64) == -2           # neither whole 32-column blocks nor one partial block
    assert call(30, 20, 64) == 0            # one partial block (F < 32)
    assert call(30, 32, 62) == -2           # rows not 16-byte aligned
    assert call(30, 64, 64, bits=False) == -2
    # a sparse batch is left to the gather by the arena
    sparse = ar.batch(random_graphs(rng, [300, 300], 0.02, True))
    assert not sparse.dense


def test_dense_routing_of_isolated_nodes():
    """A dense batch with a zero-degree node takes the CSR gather under neighbour average + learn_eps (the one mode in
    which that node's row is 0/0 = NaN, graphcnn.py:157-158, and a product would spread it over its graph from the next
    layer on); every other mode, and every batch without such a node, keeps the matrix-core kernel."""
    from gnm import core
    from gnm.arena import GraphArena
    rng = np.random.default_rng(3)
    ar = GraphArena(DEV)
    full = ar.batch(random_graphs(rng, [40, 40], 1.0, True))
    g_iso = random_graphs(rng, [40, 40], 0.5, True)
    em = g_iso[1].edge_mat.numpy()
    g_iso[1].edge_mat = torch.from_numpy(np.ascontiguousarray(em[:, (em[0] != 7) & (em[1] != 7)]))
    iso = ar.batch(g_iso)
    assert full.dense and not full.iso and iso.dense and iso.iso
    for npool in ("sum", "average"):
        for learn_eps in (True, False):
            spec = core.GinSpec(2, 2, learn_eps, "sum", npool)
            assert core._dense(full, 64, spec)
            assert core._dense(iso, 64, spec) == (not (npool == "average" and learn_eps))


@pytest.mark.gpu
def test_persistent_forward_form_opt_in():
    """GNM_AGGM_PERSIST=1 (csrc/aggm.hip, gnm_aggp_kernel: an experiment the product does not launch -- measured slower)
    still computes the same forward: the forward tests of this file once more in ONE child process with the knob set
    (the library reads it once, at its first launch)."""
    import subprocess
    import sys
    env = dict(os.environ, GNM_AGGM_PERSIST="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_aggm_forward_backward or test_aggm_forward_with_fused_bn_relu_readout"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-1500:]
