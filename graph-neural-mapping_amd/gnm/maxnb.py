"""Host side of neighbor_pooling_type == "max": what replaces __preprocess_neighbors_maxpool (graphcnn.py:55-81).

The reference builds, per forward, a [N, max_deg (+1)] LongTensor of padded neighbour lists in Python.  Here each
graph's `neighbors` lists are flattened ONCE (cached on the graph object) into a CSR in the lists' own order -- the
order decides which of several equal maxima receives the gradient -- plus the transposed, de-duplicated structure the
backward gathers over; a batch is their concatenation with the graphs' node offsets added (numpy, then one copy each).
The kernels are csrc/maxpool.hip."""
import numpy as np
import torch


class _GraphLists:
    __slots__ = ("n", "deg", "flat", "t")

    def __init__(self, graph):
        nbs = graph.neighbors
        n = len(graph.g)
        if len(nbs) != n:
            raise ValueError("graph.neighbors has %d lists for %d nodes" % (len(nbs), n))
        self.n = n
        self.deg = np.fromiter((len(x) for x in nbs), dtype=np.int64, count=n)
        self.flat = (np.concatenate([np.asarray(x, dtype=np.int64).reshape(-1) for x in nbs])
                     if n and int(self.deg.sum()) else np.zeros(0, dtype=np.int64))
        if self.flat.size and (self.flat.min() < 0 or self.flat.max() >= n):
            raise IndexError("graph.neighbors holds a node id outside [0, %d)" % n)   # the reference's gather raises too
        self.t = {}

    def transposed(self, self_last):
        """(t_deg [n], t_flat): for every node j the distinct nodes i that have j among their candidates, ascending"""
        got = self.t.get(self_last)
        if got is None:
            n = self.n
            src = np.repeat(np.arange(n, dtype=np.int64), self.deg)
            dst = self.flat
            if self_last:
                loops = np.arange(n, dtype=np.int64)
                src, dst = np.concatenate([src, loops]), np.concatenate([dst, loops])
            key = np.unique(dst * n + src)
            got = self.t[self_last] = (np.bincount(key // n, minlength=n).astype(np.int64), key % n)
        return got


def _lists_of(graph):
    cached = getattr(graph, "_gnm_maxnb", None)
    if cached is not None and cached[0] is graph.neighbors:
        return cached[1]
    gl = _GraphLists(graph)
    try:
        graph._gnm_maxnb = (graph.neighbors, gl)
    except Exception:
        pass
    return gl


class MaxNeighbours:
    """Device arrays of one batch for gnm_maxpool_fwd / gnm_maxpool_bwd (include/gnm_hip.h)."""

    def __init__(self, batch_graph, self_last, device):
        lists = [_lists_of(g) for g in batch_graph]
        ns = np.asarray([gl.n for gl in lists], dtype=np.int64)
        node_off = np.zeros(len(lists) + 1, dtype=np.int64)
        np.cumsum(ns, out=node_off[1:])
        self.N = int(node_off[-1])
        self.self_last = bool(self_last)
        self.max_deg = max(int(g.max_neighbor) for g in batch_graph)          # graphcnn.py:59 reads the field, not the lists
        deg = np.concatenate([gl.deg for gl in lists]) if lists else np.zeros(0, dtype=np.int64)
        if deg.size and int(deg.max()) > self.max_deg:
            # the reference's pad.extend([-1] * negative) leaves a longer row and torch.LongTensor raises on the ragged list
            raise ValueError("a neighbour list is longer than max_neighbor")
        col = np.concatenate([gl.flat + node_off[i] for i, gl in enumerate(lists)]) if lists else deg
        tparts = [gl.transposed(self.self_last) for gl in lists]
        t_deg = np.concatenate([t[0] for t in tparts]) if lists else deg
        t_col = np.concatenate([t[1] + node_off[i] for i, t in enumerate(tparts)]) if lists else deg
        if max(col.size, t_col.size) >= 2 ** 31:
            raise ValueError("batch has too many neighbour entries for 32-bit offsets")
        iso = np.nonzero(deg == 0)[0]

        def dev(a):
            return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(device, non_blocking=True)

        def offsets(d):
            off = np.zeros(d.size + 1, dtype=np.int64)
            np.cumsum(d, out=off[1:])
            return dev(off)

        self.nb_off, self.nb_col = offsets(deg), dev(col if col.size else np.zeros(1))
        self.t_off, self.t_col = offsets(t_deg), dev(t_col if t_col.size else np.zeros(1))
        self.n_iso = int(iso.size)
        # rows that can select the dummy; it exists for them only when some other row has neighbours (max_deg > 0)
        self.iso_rows = dev(iso) if self.n_iso and self.max_deg > 0 else None
        if self.iso_rows is None:
            self.n_iso = 0
        self.need_dummy = bool(deg.size and int(deg.min()) < self.max_deg)
