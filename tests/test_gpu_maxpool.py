"""csrc/maxpool.hip (neighbor_pooling_type == "max", graphcnn.py:55-81, 137-143) through the C-ABI against the oracle's
numpy restatement (oracle/gin_oracle.py maxpool_fwd / maxpool_bwd, pinned on the reference's goldens by
tests/test_oracle_golden.py).  Selection is index work: values and selected rows must match BIT FOR BIT, including
which of several equal maxima wins (ReLU zeros make ties the common case), NaN handling, rows without neighbours
(the dummy = column minimum) and repeated neighbours."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _G:
    pass


def _graphs(sizes, density, seed, isolate_frac=0.0, dup=False):
    rng = np.random.default_rng(seed)
    out = []
    for n in sizes:
        nb = [[] for _ in range(n)]
        for i in range(n):
            for j in range(i + 1, n):
                if rng.random() < density:
                    nb[i].append(j)
                    nb[j].append(i)
        for i in range(n):
            if rng.random() < isolate_frac:
                for j in nb[i]:
                    nb[j] = [x for x in nb[j] if x != i]
                nb[i] = []
        if dup and n > 2:
            nb[0] = nb[0] + nb[0][:2] + [0]              # repeated neighbours and a self loop in the list
        for i in range(n):
            rng.shuffle(nb[i])                           # list order is arbitrary in the reference (networkx edge order)
        g = _G()
        g.g, g.neighbors = list(range(n)), nb
        g.max_neighbor = max((len(x) for x in nb), default=0)
        out.append(g)
    return out


def _features(N, F, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "relu":                                   # what the layers above 0 see: many exact zeros
        return np.maximum(rng.standard_normal((N, F)), 0).astype(np.float32)
    if kind == "coarse":                                 # few distinct values: ties everywhere
        return rng.integers(-2, 3, (N, F)).astype(np.float32)
    h = rng.standard_normal((N, F)).astype(np.float32)
    if kind == "nan":
        h[rng.integers(0, N, max(1, N // 7)), rng.integers(0, F, max(1, N // 7))] = np.nan
    return h


def _run_fwd(mb, h, eps):
    from gnm._cabi import check, lib
    N, F = h.shape
    hd = torch.from_numpy(h).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    dummy = amin = None
    if mb.need_dummy:
        nblk = lib.gnm_maxpool_colmin_blocks(N)
        wv, wi = torch.empty(nblk, F, device=DEV), torch.empty(nblk, F, dtype=torch.int32, device=DEV)
        dummy, amin = torch.empty(F, device=DEV), torch.empty(F, dtype=torch.int32, device=DEV)
        check(lib.gnm_maxpool_colmin(hd.data_ptr(), F, N, F, wv.data_ptr(), wi.data_ptr(), dummy.data_ptr(),
                                     amin.data_ptr(), st), "colmin")
    out = torch.full((N, F + 3), float("inf"), device=DEV)               # a row stride that is not F
    amax = torch.full((N, F), -7, dtype=torch.int32, device=DEV)
    epsd = torch.tensor([eps], dtype=torch.float32, device=DEV) if eps is not None else None
    check(lib.gnm_maxpool_fwd(hd.data_ptr(), F, mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), N, F, mb.max_deg,
                              int(mb.self_last), epsd.data_ptr() if eps is not None else None,
                              dummy.data_ptr() if dummy is not None else None, out.data_ptr(), F + 3, amax.data_ptr(), st),
          "maxpool_fwd")
    assert torch.isinf(out[:, F:]).all()
    return out[:, :F].cpu().numpy(), amax, amin, dummy


CASES = [([20, 20, 20], 0.3, 32, "relu", 0.0, False), ([7, 1, 13], 0.5, 7, "normal", 0.2, False),
         ([40], 0.9, 64, "coarse", 0.0, True), ([12, 12], 0.2, 5, "coarse", 0.3, True), ([30, 30], 0.4, 64, "nan", 0.1, False),
         ([3] * 50, 1.0, 64, "relu", 0.0, False), ([64], 0.05, 33, "relu", 0.5, False)]


@pytest.mark.parametrize("learn_eps", [True, False])
@pytest.mark.parametrize("sizes,density,F,kind,iso,dup", CASES)
def test_maxpool_forward_and_backward_bit_exact(sizes, density, F, kind, iso, dup, learn_eps):
    from gnm._cabi import check, lib
    from gnm.maxnb import MaxNeighbours
    from oracle import gin_oracle as O
    graphs = _graphs(sizes, density, seed=sum(sizes) + F, isolate_frac=iso, dup=dup)
    mb = MaxNeighbours(graphs, not learn_eps, DEV)
    if mb.max_deg == 0 and learn_eps:
        pytest.skip("no candidates at all: covered by test_maxpool_refusals")
    N = mb.N
    h = _features(N, F, seed=N + F, kind=kind)
    for g in graphs:
        g.num_nodes = len(g.g)
    padded = O.build_padded_neighbors(graphs, learn_eps)
    want, cache = O.maxpool_fwd(h, padded)
    eps = 0.37 if learn_eps else None
    if learn_eps:
        want = want + (np.float32(1) + np.float32(eps)) * h                # fp32, op by op (graphcnn.py:161)
    got, amax, amin, dummy = _run_fwd(mb, h, eps)
    assert np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(amax.cpu().numpy(), cache["src"])                # which candidate won, ties and NaN included
    if mb.need_dummy:
        assert np.array_equal(amin.cpu().numpy(), cache["amin"])
        assert np.array_equal(dummy.cpu().numpy(), h[cache["amin"], np.arange(F)], equal_nan=True)
    # backward: integer-valued gradients make every summation order exact -> bit-exact too
    rng = np.random.default_rng(5)
    g = rng.integers(-3, 4, (N, F)).astype(np.float32)
    want_dh = O.maxpool_bwd(g.copy(), cache)
    epsd = None
    if learn_eps:
        eps = 1.0                                                           # (1 + eps) = 2: still exact
        want_dh = want_dh + np.float32(2) * g
        epsd = torch.tensor([eps], dtype=torch.float32, device=DEV)
    gd = torch.from_numpy(g).to(DEV)
    dh = torch.full((N, F), float("nan"), device=DEV)
    check(lib.gnm_maxpool_bwd(gd.data_ptr(), F, amax.data_ptr(), mb.t_off.data_ptr(), mb.t_col.data_ptr(), N, F,
                              epsd.data_ptr() if epsd is not None else None,
                              mb.iso_rows.data_ptr() if mb.n_iso else None, mb.n_iso,
                              amin.data_ptr() if amin is not None else None, dh.data_ptr(), F,
                              torch.cuda.current_stream().cuda_stream), "maxpool_bwd")
    if kind != "nan":
        assert np.array_equal(dh.cpu().numpy(), want_dh)
    _check_tiled(graphs, mb, h, 0.37 if learn_eps else None, got, amax, amin, dummy, gd, epsd, dh)


def _check_tiled(graphs, mb, h, eps, got, amax, amin, dummy, gd, epsd, dh):
    """the one-workgroup-per-graph forms (rows in LDS) take F = 32 / 64 and give the bits of the gather-from-L2 forms;
    other shapes are declined"""
    from gnm._cabi import lib
    N, F = h.shape
    sizes = [len(g.g) for g in graphs]
    node_off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=DEV)
    hd = torch.from_numpy(h).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    out2 = torch.full((N, F + 4), float("inf"), device=DEV)
    amax2 = torch.full((N, F), -7, dtype=torch.int32, device=DEV)
    e1 = torch.tensor([eps], dtype=torch.float32, device=DEV) if eps is not None else None
    rc = lib.gnm_maxpool_fwd_tiled(hd.data_ptr(), F, mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), node_off.data_ptr(),
                                   len(sizes), max(sizes), F, mb.max_deg, int(mb.self_last),
                                   e1.data_ptr() if e1 is not None else None,
                                   dummy.data_ptr() if dummy is not None else None, out2.data_ptr(), F + 4,
                                   amax2.data_ptr(), st)
    assert rc == (0 if F in (32, 64) else -2)
    dh2 = torch.full((N, F), float("nan"), device=DEV)
    rc2 = lib.gnm_maxpool_bwd_tiled(gd.data_ptr(), F, amax.data_ptr(), mb.t_off.data_ptr(), mb.t_col.data_ptr(),
                                    node_off.data_ptr(), len(sizes), max(sizes), F,
                                    epsd.data_ptr() if epsd is not None else None,
                                    mb.iso_rows.data_ptr() if mb.n_iso else None, mb.n_iso,
                                    amin.data_ptr() if amin is not None else None, dh2.data_ptr(), F, st)
    assert rc2 == rc
    if rc == 0:
        assert np.array_equal(out2[:, :F].cpu().numpy(), got, equal_nan=True) and torch.isinf(out2[:, F:]).all()
        assert torch.equal(amax2, amax)
        assert torch.equal(dh2, dh) or bool(torch.isnan(dh).any())


def test_maxpool_refusals_and_empty():
    from gnm._cabi import lib
    from gnm.maxnb import MaxNeighbours
    g = _G()
    g.g, g.neighbors, g.max_neighbor = [0, 1], [[], []], 0
    mb = MaxNeighbours([g], False, DEV)
    h = torch.zeros(2, 4, device=DEV)
    out = torch.empty(2, 4, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    # no candidate at all (torch.max over an empty dimension raises): refused
    assert lib.gnm_maxpool_fwd(h.data_ptr(), 4, mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), 2, 4, 0, 0, None, None,
                               out.data_ptr(), 4, None, st) == -1
    # the self-loop form of the same graphs: every row is its own maximum
    assert lib.gnm_maxpool_fwd(h.data_ptr(), 4, mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), 2, 4, 0, 1, None, None,
                               out.data_ptr(), 4, None, st) == 0
    assert lib.gnm_maxpool_fwd(h.data_ptr(), 4, mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), 0, 4, 0, 0, None, None,
                               out.data_ptr(), 4, None, st) == 0          # N = 0: nothing to do
    assert lib.gnm_maxpool_fwd(h.data_ptr(), 3, mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), 2, 4, 0, 1, None, None,
                               out.data_ptr(), 4, None, st) == -1         # row stride shorter than a row
    assert lib.gnm_maxpool_colmin(None, 4, 2, 4, None, None, None, None, st) == -1


def test_maxpool_true_shape_properties_and_determinism():
    """64 graphs of 400 nodes at the headline density, F = 64: the pooled value is attained by a listed neighbour,
    no neighbour exceeds it, the backward conserves the gradient mass, and two runs agree bitwise."""
    from gnm import synth
    from gnm._cabi import check, lib
    from gnm.maxnb import MaxNeighbours
    graphs = [synth.dense_fc_graph(s, n=400, f0=1) for s in range(64)]
    for g in graphs:
        em = g.edge_mat.numpy()
        g.neighbors = [[] for _ in range(400)]
        for i, j in em.T:
            g.neighbors[int(i)].append(int(j))
        g.max_neighbor = max(len(x) for x in g.neighbors)
    mb = MaxNeighbours(graphs, True, DEV)                                      # the self-loop form (learn_eps False)
    N, F = mb.N, 64
    h = _features(N, F, 3, "relu")
    got, amax, amin, dummy = _run_fwd(mb, h, None)
    got2, amax2, _, _ = _run_fwd(mb, h, None)
    assert np.array_equal(got, got2) and torch.equal(amax, amax2)
    # ... and the tiled form (what the model runs at this shape) gives the same bits, forward and backward
    node_off = torch.arange(65, dtype=torch.int32, device=DEV) * 400
    hd, st0 = torch.from_numpy(h).to(DEV), torch.cuda.current_stream().cuda_stream
    out3, amax3 = torch.empty(N, F, device=DEV), torch.empty(N, F, dtype=torch.int32, device=DEV)
    check(lib.gnm_maxpool_fwd_tiled(hd.data_ptr(), F, mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), node_off.data_ptr(), 64,
                                    400, F, mb.max_deg, 1, None, dummy.data_ptr() if dummy is not None else None,
                                    out3.data_ptr(), F, amax3.data_ptr(), st0), "fwd_tiled")
    assert np.array_equal(out3.cpu().numpy(), got) and torch.equal(amax3, amax)
    gr = torch.randn(N, F, device=DEV)
    d1, d2 = torch.empty(N, F, device=DEV), torch.empty(N, F, device=DEV)
    epsd = torch.tensor([0.25], device=DEV)
    check(lib.gnm_maxpool_bwd(gr.data_ptr(), F, amax.data_ptr(), mb.t_off.data_ptr(), mb.t_col.data_ptr(), N, F,
                              epsd.data_ptr(), None, 0, None, d1.data_ptr(), F, st0), "bwd")
    check(lib.gnm_maxpool_bwd_tiled(gr.data_ptr(), F, amax.data_ptr(), mb.t_off.data_ptr(), mb.t_col.data_ptr(),
                                    node_off.data_ptr(), 64, 400, F, epsd.data_ptr(), None, 0, None, d2.data_ptr(), F, st0),
          "bwd_tiled")
    assert torch.equal(d1, d2)
    am = amax.cpu().numpy()
    assert (am >= 0).all()                                                     # every row has its self candidate
    assert np.array_equal(got, h[am, np.arange(F)[None, :]])                   # attained by the selected row
    off, col = mb.nb_off.cpu().numpy(), mb.nb_col.cpu().numpy()
    rows = np.repeat(np.arange(N), np.diff(off))
    assert (h[col] <= got[rows]).all() and (h <= got).all()                    # nothing listed exceeds it
    g = torch.ones(N, F, device=DEV)
    dh = torch.empty(N, F, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        check(lib.gnm_maxpool_bwd(g.data_ptr(), F, amax.data_ptr(), mb.t_off.data_ptr(), mb.t_col.data_ptr(), N, F, None,
                                  None, 0, None, dh.data_ptr(), F, st), "bwd")
    assert float(dh.sum()) == float(N * F)                                     # each element's gradient lands exactly once
    cnt = np.zeros((N, F), dtype=np.int64)
    np.add.at(cnt, (am, np.broadcast_to(np.arange(F), am.shape)), 1)
    assert np.array_equal(dh.cpu().numpy(), cnt.astype(np.float32))
