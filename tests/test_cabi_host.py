"""CPU-only checks of the C-ABI library and the host logic above it: the .so loads and
exports every symbol include/gnm_hip.h declares (no device compute is called), the host
CSR helpers reproduce the reference's index structures bit-exactly, the module mirrors
the reference's constructor / state_dict / seeded init, and the product path fails loudly
without a GPU instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import edge_mat_of, golden_cases, load_case, neighbors_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gnm import _cabi
    header = open(os.path.join(ROOT, "include", "gnm_hip.h")).read()
    declared = set(re.findall(r"\b(gnm_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(_cabi.lib, name), name
    assert declared == set(_cabi.SIGNATURES), declared ^ set(_cabi.SIGNATURES)
    assert _cabi.lib.gnm_version().startswith(b"gnm_hip")


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "graph-neural-mapping_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, re.M), f
                assert not re.search(r"^\s*(import|from)\s+\S*gin_oracle\b", txt, re.M), f
                assert "oracle/" not in txt.replace("oracle/gin_oracle.py, which the tests", ""), f


class G:
    pass


def graphs_of(cfg, d):
    out = []
    for g in range(cfg["B"]):
        o = G()
        o.g = list(range(cfg["n"]))
        o.edge_mat = torch.from_numpy(edge_mat_of(d[f"und_{g}"]))
        o.node_features = torch.from_numpy(d[f"feat_{g}"])
        o.label = int(d["labels"][g])
        o.neighbors = neighbors_of(d[f"und_{g}"], cfg["n"])
        o.max_neighbor = max(len(x) for x in o.neighbors)
        out.append(o)
    return out


@pytest.mark.parametrize("case", golden_cases())
def test_csr_arena_matches_reference_coo_bit_exact(case):
    """CSR (host build, CPU arena) expands back to exactly Adj_block._indices()
    (graphcnn.py:91-104) as an edge multiset, offsets included."""
    from gnm.arena import GraphArena
    cfg, state, d = load_case(case)
    ar = GraphArena("cpu")
    batch = ar.batch(graphs_of(cfg, d))
    assert batch.symmetric and batch.N == cfg["B"] * cfg["n"] and batch.equal_n
    if cfg["npool"] == "max":
        _check_max_neighbour_structure(cfg, d, graphs_of(cfg, d))
    else:
        coo = ar.export_adj_coo(batch, self_loops=not cfg["learn_eps"])
        ref = d["adj_indices"]
        canon = lambda a: a[:, np.lexsort((a[1], a[0]))]
        assert coo.dtype == np.int64 and np.array_equal(canon(coo), canon(ref))
    X = ar.features(batch).numpy()
    assert np.array_equal(X, np.concatenate([d[f"feat_{g}"] for g in range(cfg["B"])]))
    # graphs are cached on the object: a second batch adds nothing
    n_before = len(ar)
    ar.batch(graphs_of(cfg, d)[:1])
    assert len(ar) == n_before + 1          # new python objects -> new entries
    gs = graphs_of(cfg, d)
    ar.batch(gs); k = len(ar); ar.batch(gs)
    assert len(ar) == k


def _check_max_neighbour_structure(cfg, d, graphs):
    """gnm.maxnb.MaxNeighbours holds exactly the reference's padded neighbour list (graphcnn.py:55-81): expanding its
    CSR the way the kernel walks it (neighbours, one dummy per short row standing for all its -1 slots, the node
    itself in the self-loop form) gives the golden LongTensor; the transposed structure lists, for every row, the
    distinct rows that have it as a candidate."""
    from gnm.maxnb import MaxNeighbours
    ref = d["padded_neighbors"]
    mb = MaxNeighbours(graphs, not cfg["learn_eps"], "cpu")
    off, col = mb.nb_off.numpy(), mb.nb_col.numpy()
    assert mb.N == ref.shape[0] and mb.max_deg == ref.shape[1] - (0 if cfg["learn_eps"] else 1)
    rows = []
    for i in range(mb.N):
        r = list(col[off[i]:off[i + 1]])
        r += [-1] * (mb.max_deg - len(r))
        if mb.self_last:
            r.append(i)
        rows.append(r)
    assert np.array_equal(np.asarray(rows, dtype=np.int64).reshape(ref.shape), ref)
    deg = np.diff(off)
    assert mb.need_dummy == bool((deg < mb.max_deg).any())
    assert mb.n_iso == int((deg == 0).sum()) and (mb.n_iso == 0 or np.array_equal(mb.iso_rows.numpy(), np.nonzero(deg == 0)[0]))
    toff, tcol = mb.t_off.numpy(), mb.t_col.numpy()
    for j in range(mb.N):
        want = sorted({i for i in range(mb.N) if j in ref[i]})
        assert list(tcol[toff[j]:toff[j + 1]]) == want


def test_max_neighbour_lists_are_cached_and_validated():
    from gnm.maxnb import MaxNeighbours
    cfg, state, d = load_case("tiny_s1_eps1_gsum_nmax")
    gs = graphs_of(cfg, d)
    MaxNeighbours(gs, False, "cpu")
    cached = gs[0]._gnm_maxnb[1]
    MaxNeighbours(gs[:2], True, "cpu")
    assert gs[0]._gnm_maxnb[1] is cached and set(cached.t) == {False, True}
    gs[0].neighbors = [list(x) for x in gs[0].neighbors]           # a new list object: re-read
    MaxNeighbours(gs[:1], False, "cpu")
    assert gs[0]._gnm_maxnb[1] is not cached
    gs[1].neighbors[0] = gs[1].neighbors[0] + [cfg["n"]]            # id outside the graph (the reference's gather raises)
    gs[1].neighbors = list(gs[1].neighbors)
    with pytest.raises(IndexError):
        MaxNeighbours(gs[1:2], False, "cpu")
    gs[2].max_neighbor = 1                                           # shorter than a list: torch.LongTensor(ragged) raises
    with pytest.raises(ValueError):
        MaxNeighbours(gs[2:3], False, "cpu")
    # duplicates and self loops in the lists: one transposed entry per distinct (candidate, row) pair
    o = G()
    o.g, o.neighbors, o.max_neighbor = [0, 1, 2], [[1, 1, 0], [0], []], 3
    mb = MaxNeighbours([o], True, "cpu")
    assert mb.t_off.tolist() == [0, 2, 4, 5] and mb.t_col.tolist() == [0, 1, 0, 1, 2]
    assert mb.iso_rows.tolist() == [2] and mb.need_dummy


def test_csr_edge_cases():
    from gnm._cabi import lib
    # empty graph
    rp = np.zeros(4, dtype=np.int32); col = np.zeros(1, dtype=np.uint16)
    em = np.zeros((2, 0), dtype=np.int64)
    assert lib.gnm_csr_from_edge_mat(em.ctypes.data, 0, 3, rp.ctypes.data, col.ctypes.data) == 0
    assert rp.tolist() == [0, 0, 0, 0]
    # duplicates kept, order within a row stable, asymmetric detected
    em = np.ascontiguousarray(np.array([[0, 0, 2, 0], [1, 2, 1, 1]], dtype=np.int64))
    rp = np.zeros(4, dtype=np.int32); col = np.zeros(4, dtype=np.uint16)
    assert lib.gnm_csr_from_edge_mat(em.ctypes.data, 4, 3, rp.ctypes.data, col.ctypes.data) == 0
    assert rp.tolist() == [0, 3, 3, 4] and col.tolist() == [1, 2, 1, 1]
    assert lib.gnm_csr_is_symmetric(rp.ctypes.data, col.ctypes.data, 3) == 0
    rpt = np.zeros(4, dtype=np.int32); ct = np.zeros(4, dtype=np.uint16)
    lib.gnm_csr_transpose(rp.ctypes.data, col.ctypes.data, 3, rpt.ctypes.data, ct.ctypes.data)
    assert rpt.tolist() == [0, 0, 3, 4] and ct.tolist() == [0, 0, 2, 0]
    # out-of-range node id is an error, not a crash
    bad = np.ascontiguousarray(np.array([[0], [5]], dtype=np.int64))
    assert lib.gnm_csr_from_edge_mat(bad.ctypes.data, 1, 3, rp.ctypes.data, col.ctypes.data) < 0
    # slice-width planning: the n=400/F=64 tile is one 100-KiB slice, C4 (n=1000,F=128) is 4 x 32
    assert lib.gnm_agg_slice_width(64, 400) == 64
    assert lib.gnm_agg_slice_width(7, 400) == 8
    assert lib.gnm_agg_slice_width(128, 1000) == 32
    assert lib.gnm_agg_slice_width(400, 400) == 64
    assert lib.gnm_agg_slice_width(64, 60000) == 0


def test_module_mirrors_reference_interface():
    from models.graphcnn import GIN_InfoMaxReg, GraphCNN, MLP, Discriminator
    assert GraphCNN is GIN_InfoMaxReg
    cfg, state, d = load_case("tiny_s0_eps1_gsum_nsum")
    torch.manual_seed(0)
    m = GIN_InfoMaxReg(cfg["L"], cfg["m"], cfg["f0"], cfg["H"], cfg["C"], 0.5, True, "sum", "sum", torch.device("cpu"))
    sd = m.state_dict()
    assert list(sd.keys()) == list(state.keys())
    perturbed = lambda k: k == "eps" or ("batch_norms" in k and (k.endswith(".weight") or k.endswith(".bias")))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(np.asarray(state[k]).shape), k
        if not perturbed(k):      # make_goldens perturbs eps / BN affine after the seeded init
            assert np.array_equal(sd[k].numpy(), state[k]), k
    with pytest.raises(ValueError):
        MLP(0, 4, 4, 4)                                       # mlp.py:21-22
    assert list(MLP(1, 4, 8, 3).state_dict().keys()) == ["linear.weight", "linear.bias"]
    assert list(Discriminator(6).state_dict().keys()) == ["f_k.weight", "f_k.bias"]


def test_no_cpu_fallback():
    from gnm._cabi import GnmError
    from models.graphcnn import GIN_InfoMaxReg
    cfg, state, d = load_case("tiny_s1_eps1_gsum_nsum")
    m = GIN_InfoMaxReg(cfg["L"], cfg["m"], cfg["f0"], cfg["H"], cfg["C"], 0.0, True, "sum", "sum", torch.device("cpu"))
    with pytest.raises(GnmError):
        m(graphs_of(cfg, d))
    with pytest.raises(AssertionError):
        m.compute_saliency(graphs_of(cfg, d), 0)              # B must be 1 (graphcnn.py:257)


def test_max_pooling_has_no_cpu_fallback_either():
    """neighbor_pooling_type='max' runs through the same HIP engine (csrc/maxpool.hip): on a CPU device it raises."""
    from gnm._cabi import GnmError
    from models.graphcnn import GIN_InfoMaxReg
    cfg, state, d = load_case("tiny_s1_eps1_gsum_nmax")
    gs = graphs_of(cfg, d)
    m = GIN_InfoMaxReg(cfg["L"], cfg["m"], cfg["f0"], cfg["H"], cfg["C"], 0.0, True, "sum", "max", torch.device("cpu"))
    with pytest.raises(GnmError):
        m(gs)
    with pytest.raises(GnmError):
        m.compute_saliency(gs[:1], 1)

def test_csr_parity_order_keeps_the_multiset_and_alternates_parities():
    """gnm_csr_parity_order (round 4): rows reordered for the 128-byte-row gather -- same ids per row, position j even
    for (j & 3) < 2 and odd otherwise for as long as the row still has both kinds, order inside a parity class kept."""
    from gnm._cabi import lib
    rng = np.random.default_rng(3)
    n = 200
    deg = rng.integers(0, 70, n)
    deg[:3] = [0, 1, 2]
    rowptr = np.zeros(n + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n, int(rowptr[-1])).astype(np.uint16)
    col[rowptr[10]:rowptr[11]] &= 0xFFFE                       # a row of even ids only
    before = col.copy()
    assert lib.gnm_csr_parity_order(rowptr.ctypes.data, col.ctypes.data, n) == 0
    for r in range(n):
        a, b = before[rowptr[r]:rowptr[r + 1]], col[rowptr[r]:rowptr[r + 1]]
        assert sorted(a.tolist()) == sorted(b.tolist())
        assert b[b % 2 == 0].tolist() == a[a % 2 == 0].tolist() and b[b % 2 == 1].tolist() == a[a % 2 == 1].tolist()
        ne, no = int((a % 2 == 0).sum()), int((a % 2 == 1).sum())
        used_e = used_o = 0
        for j, v in enumerate(b.tolist()):
            want_odd = (j & 3) >= 2
            if (want_odd and used_o < no) or (not want_odd and used_e >= ne):
                assert v % 2 == 1, (r, j)
                used_o += 1
            else:
                assert v % 2 == 0, (r, j)
                used_e += 1
