"""Synthetic connectivity graphs of the shapes BASELINE.json names.

dense_fc: what the reference's data path produces for one subject with the default
--sparsity 30 (/root/reference dataset.py:93-101 via util.py:37, main.py:105): a 400x400
correlation matrix thresholded at its 70th percentile over the WHOLE matrix (diagonal
included), upper triangle kept, both directions emitted (util.py:99-100)
=> exactly 23,800 undirected = 47,600 directed edges at n = 400.
knn: the sparse regime (config 4): k nearest neighbours in a random 8-d embedding,
symmetrised by union.
"""
import numpy as np
import torch


class SynthGraph:
    """S2VGraph-shaped object (util.py:9-17): g, label, edge_mat, node_features."""

    __slots__ = ("g", "label", "edge_mat", "node_features", "neighbors", "max_neighbor", "node_tags", "_gnm_cache",
                 "_gnm_maxnb")

    def __init__(self, n, und_edges, feats, label):
        self.g = range(n)                                   # only len(graph.g) is read
        self.label = int(label)
        e = np.asarray(und_edges, dtype=np.int64).reshape(-1, 2)
        self.edge_mat = torch.from_numpy(np.ascontiguousarray(np.concatenate([e, e[:, ::-1]], 0).T))
        self.node_features = torch.from_numpy(np.ascontiguousarray(feats, dtype=np.float32))
        self.neighbors = None
        self.max_neighbor = 0
        self.node_tags = None
        self._gnm_cache = None
        self._gnm_maxnb = None

    def build_neighbors(self):
        """graph.neighbors / max_neighbor as util.py:86-95 fills them (read by "max" neighbour pooling only): both ends of
        every undirected edge in edge order -- here from edge_mat, whose first half lists (i, j) and second half (j, i)"""
        em = self.edge_mat.numpy()
        n = len(self.g)
        half = em.shape[1] // 2
        src = np.stack([em[0, :half], em[1, :half]], 1).reshape(-1)       # i0, j0, i1, j1, ...
        dst = np.stack([em[1, :half], em[0, :half]], 1).reshape(-1)
        order = np.argsort(src, kind="stable")
        cnt = np.bincount(src, minlength=n)
        self.neighbors = [x.tolist() for x in np.split(dst[order], np.cumsum(cnt)[:-1])]
        self.max_neighbor = int(cnt.max()) if n else 0
        return self


def dense_fc_graph(g, n=400, t=256, f0=7, keep_pct=30.0):
    rng = np.random.default_rng(1000 + g)
    ts = rng.standard_normal((t, n))
    cm = np.corrcoef(ts.T)
    mask = cm > np.percentile(cm, 100.0 - keep_pct)
    iu, ju = np.nonzero(np.triu(mask, 1))
    feats = rng.standard_normal((n, f0)).astype(np.float32)
    label = int(rng.integers(0, 2))
    return SynthGraph(n, np.stack([iu, ju], 1), feats, label)


def knn_graph(g, n=1000, k=20, dim=8, f0=7):
    rng = np.random.default_rng(5000 + g)
    P = rng.standard_normal((n, dim))
    d2 = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    nb = np.argpartition(d2, k, axis=1)[:, :k]
    A = np.zeros((n, n), dtype=bool)
    A[np.repeat(np.arange(n), k), nb.reshape(-1)] = True
    A |= A.T                                                # union symmetrisation
    iu, ju = np.nonzero(np.triu(A, 1))
    feats = rng.standard_normal((n, f0)).astype(np.float32)
    return SynthGraph(n, np.stack([iu, ju], 1), feats, int(rng.integers(0, 2)))


def make_pool(kind, count, first=0, threads=None, **kw):
    """count graphs with ids first .. first+count-1 (each graph is a pure function of its id).  Generation is
    numpy-bound (corrcoef, percentile sort) and releases the GIL, so large pools are built on a few threads."""
    fn = dense_fc_graph if kind == "dense_fc" else knn_graph
    if threads is None:
        import os
        threads = min(16, os.cpu_count() or 1)
    if count < 64 or threads <= 1:
        return [fn(first + i, **kw) for i in range(count)]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(lambda i: fn(first + i, **kw), range(count)))
