"""Device-resident graph arena and batch descriptors.

Replaces the per-forward host work of GIN_InfoMaxReg.__preprocess_neighbors_sumavepool
(/root/reference models/graphcnn.py:84-106: int64 block-diagonal COO rebuilt and copied
H2D every forward, 780 MB at B=1024), __preprocess_graphpool (:109-134, O(N) Python
loops) and the X_concat H2D copy (:195).  Each S2VGraph-like object is converted ONCE to
a graph-local CSR (int32 rowptr, uint16 columns) and uploaded with its node features;
a batch is then just B graph ids, from which three small offset vectors are gathered on
the device.
"""

import os

import numpy as np
import torch

from ._cabi import GnmError, check, lib

# A batch takes the matrix-core aggregation (csrc/aggm.hip: bit adjacency x bf16 planes) when every graph in it has
# a bit matrix and the batch is at least this dense (edges / nodes^2): the product costs ~n^2 per graph whatever the
# edge count, the gather ~edges plus a fixed part.  Measured on MI355X (tools/bench_agg.py --density, B = 1024,
# n = 400, F = 64, us per launch, gather / matrix-core): 4 % 89 / 82, 8 % 97 / 80, 12 % 111 / 84, 20 % 153 / 82, the
# benchmark's 30 % 175 / 83 -- the product already wins at the sparsest point measured, which is the threshold.
# Tuning knob, read once: GNM_DENSE_FILL (a value > 1 turns the matrix-core path off).
DENSE_MIN_FILL = float(os.environ.get("GNM_DENSE_FILL", "0.04"))
# GNM_NO_PARITY_ORDER=1: keep every CSR row in edge_mat order (A/B of the id ordering for the 128-byte-row gather)
PARITY_ORDER = os.environ.get("GNM_NO_PARITY_ORDER") is None


class _Growable:
    """1-D device buffer with amortised doubling growth."""

    def __init__(self, dtype, device, width=None, slack=0):
        self.dtype, self.device, self.width = dtype, device, width
        self.slack = slack            # elements kept allocated (and zeroed) past `size`
        self.size = 0
        shape = (0,) if width is None else (0, width)
        self.buf = torch.empty(shape, dtype=dtype, device=device)

    def append(self, host_tensor):
        k = host_tensor.shape[0]
        need = self.size + k
        if need + self.slack > self.buf.shape[0]:
            cap = max(need + self.slack, 2 * self.buf.shape[0], 1024)
            shape = (cap,) if self.width is None else (cap, self.width)
            nb = torch.zeros(shape, dtype=self.dtype, device=self.device)
            nb[: self.size].copy_(self.buf[: self.size])
            self.buf = nb
        self.buf[self.size:need].copy_(host_tensor, non_blocking=False)
        off = self.size
        self.size = need
        return off

    def reserve(self, k):
        """k zeroed elements at the end (filled on the device by the caller); returns their offset"""
        need = self.size + k
        if need + self.slack > self.buf.shape[0]:
            cap = max(need + self.slack, 2 * self.buf.shape[0], 1024)
            nb = torch.zeros((cap,) if self.width is None else (cap, self.width), dtype=self.dtype, device=self.device)
            nb[: self.size].copy_(self.buf[: self.size])
            self.buf = nb
        else:
            self.buf[self.size:need].zero_()
        off = self.size
        self.size = need
        return off


class Batch:
    """What the kernels need to know about one batch of graphs (all on the device)."""

    __slots__ = ("B", "N", "n_max", "n_min", "nnz_max", "arena", "node_off", "rp_off", "col_off", "t_rp_off",
                 "t_col_off", "gids", "node_off_host", "symmetric", "feat_base", "bits_off", "t_bits_off", "dense", "maxnb", "iso", "has_bits")
    # has_bits: every graph of the batch has a bit adjacency (whatever the batch's density): the one-launch eval encoder
    # iso: some graph of the batch has a node without neighbours.  Under neighbour "average" + learn_eps that node's
    # row is 0/0 = NaN (graphcnn.py:157-158) and the reference keeps the NaN to ITS row and its neighbours' -- none -- per
    # layer; the matrix-core product would spread it to every row of the graph (0 x NaN), so such batches take the
    # CSR gather in that mode (gnm/core.py _dense)
    # maxnb: gnm.maxnb.MaxNeighbours of the batch when the model pools neighbours with "max" (set by forward())
    # dense: every graph has a bit adjacency and the batch is dense enough for the matrix-core aggregation
    # (DENSE_MIN_FILL); bits_off / t_bits_off: int64 [B] offsets of the forward / transposed bit matrices in arena.bits
    # feat_base: int64 [B], first row of each graph's node features in the arena's feature buffer.  Part of the
    # batch (not looked up in the arena's tables at use): a captured hipGraph replays the lookup's kernels with the
    # table tensor of capture time, which is freed as soon as another graph is added to the arena.

    @property
    def equal_n(self):
        return self.n_max == self.n_min


class GraphArena:
    def __init__(self, device):
        self.device = torch.device(device)
        self.rowptr = _Growable(torch.int32, self.device)
        # uint16 payload; the gather kernels fetch column ids 128 at a time without bounds
        # branches, so 128 readable ids are kept past the last block (include/gnm_hip.h)
        self.col = _Growable(torch.int16, self.device, slack=256)
        self.feat = None                                    # created on first add (needs F0)
        # bit adjacency of every graph small enough for csrc/aggm.hip (built on the device from the CSR at add time)
        self.bits = _Growable(torch.int32, self.device)
        self.bits_off, self.t_bits_off, self.bits_ok = [], [], []
        self.n, self.rp_off, self.col_off, self.t_rp_off, self.t_col_off, self.feat_off, self.nnz = [], [], [], [], [], [], []
        self.sym = []
        self.iso = []               # per graph: has a node with no neighbours (zero-length CSR row)
        self._dev_tables = None
        self._token = object()
        self._agg0 = {}
        self._minus_one = None
        self._arange = {}           # {n: arange(n) on the device}: one launch less per feature gather

    def __len__(self):
        return len(self.n)

    # ------------------------------------------------------------------ build
    def add(self, graph):
        """Convert one S2VGraph-like object (fields: g, edge_mat, node_features;
        util.py:9-17) and return its arena id.  Cached on the object."""
        cached = getattr(graph, "_gnm_cache", None)
        if cached is not None and cached[0] is self._token:
            return cached[1]
        n = len(graph.g)
        em = graph.edge_mat
        em = em.detach().cpu().numpy() if torch.is_tensor(em) else np.asarray(em)
        em = np.ascontiguousarray(em, dtype=np.int64).reshape(2, -1)
        feats = graph.node_features
        feats = feats.detach().cpu() if torch.is_tensor(feats) else torch.as_tensor(np.asarray(feats))
        feats = feats.to(torch.float32).contiguous()
        gid = self.add_raw(n, em, feats)
        try:
            graph._gnm_cache = (self._token, gid)
        except Exception:
            pass
        return gid

    @staticmethod
    def _host_csr(n, edge_mat):
        """(rowptr int32 [n+1], col uint16 [E], transposed pair or None) of one graph, on the host (C calls that
        release the GIL)."""
        E = edge_mat.shape[1]
        rowptr = np.empty(n + 1, dtype=np.int32)
        col = np.empty(max(E, 1), dtype=np.uint16)
        check(lib.gnm_csr_from_edge_mat(edge_mat.ctypes.data, E, n, rowptr.ctypes.data, col.ctypes.data),
              "gnm_csr_from_edge_mat")
        col = col[:E]
        sym = bool(lib.gnm_csr_is_symmetric(rowptr.ctypes.data, col.ctypes.data if E else None, n)) if E else True
        tr = None
        if not sym:
            rpt = np.empty(n + 1, dtype=np.int32)
            ct = np.empty(max(E, 1), dtype=np.uint16)
            check(lib.gnm_csr_transpose(rowptr.ctypes.data, col.ctypes.data, n, rpt.ctypes.data, ct.ctypes.data),
                  "gnm_csr_transpose")
            ct = ct[:E]
            if PARITY_ORDER:
                check(lib.gnm_csr_parity_order(rpt.ctypes.data, ct.ctypes.data, n), "gnm_csr_parity_order")
            tr = (rpt, ct)
        # rows ordered by id parity for the 128-byte-row gather (include/gnm_hip.h gnm_csr_parity_order): same multiset
        if E and PARITY_ORDER:
            check(lib.gnm_csr_parity_order(rowptr.ctypes.data, col.ctypes.data, n), "gnm_csr_parity_order")
        return rowptr, col, tr

    def add_raw(self, n, edge_mat, feats):
        return self._append_host([(n, self._host_csr(n, edge_mat), feats)])[0]

    def _append_host(self, items):
        """items: [(n, (rowptr, col, transposed or None), feats)].  ONE upload per array for the whole list
        (round 1 made three synchronous copies per graph); returns the new arena ids."""
        if not items:
            return []
        rps, cols, fts = [], [], []
        rp_pos, col_pos, feat_pos = self.rowptr.size, self.col.size, (self.feat.size if self.feat is not None else 0)
        F0 = items[0][2].shape[1] if self.feat is None else self.feat.width
        meta = []
        for n, (rowptr, col, tr), feats in items:
            if tuple(feats.shape) != (n, F0):
                raise ValueError("node_features must be [n, %d], got %s" % (F0, tuple(feats.shape)))
            E = int(col.shape[0])
            rp_off, col_off = rp_pos, col_pos
            rps.append(rowptr); cols.append(col)
            rp_pos += n + 1; col_pos += E
            if tr is None:
                t_rp_off, t_col_off = rp_off, col_off
            else:
                t_rp_off, t_col_off = rp_pos, col_pos
                rps.append(tr[0]); cols.append(tr[1])
                rp_pos += n + 1; col_pos += E
            fts.append(feats)
            iso = bool(n > 0 and (np.diff(rowptr) == 0).any())
            meta.append((n, E, tr is None, rp_off, col_off, t_rp_off, t_col_off, feat_pos, iso))
            feat_pos += n
        if self.feat is None:
            self.feat = _Growable(torch.float32, self.device, width=F0)
        off = self.rowptr.append(torch.from_numpy(np.concatenate(rps)))
        assert off == meta[0][3]
        self.col.append(torch.from_numpy(np.concatenate(cols).view(np.int16)))
        self.feat.append(torch.cat(fts, 0) if len(fts) > 1 else fts[0])
        first = len(self.n)
        for n, E, sym, rp_off, col_off, t_rp_off, t_col_off, feat_off, iso in meta:
            self.n.append(n); self.nnz.append(E); self.sym.append(sym); self.iso.append(iso)
            self.rp_off.append(rp_off); self.col_off.append(col_off)
            self.t_rp_off.append(t_rp_off); self.t_col_off.append(t_col_off); self.feat_off.append(feat_off)
        self._build_bits(first)
        self._dev_tables = None
        return list(range(first, len(self.n)))

    def _build_bits(self, first):
        """bit adjacency (include/gnm_hip.h, gnm_adj_bits_build) of the graphs first.. : forward, and transposed where
        the graph is not symmetric.  Graphs too large for the matrix-core kernel, multigraphs (a repeated edge: the
        bit cannot carry its weight) and arenas that are not on a GPU get none (bits_ok False -> CSR gather)."""
        G = len(self.n)
        self.bits_off += [0] * (G - first)
        self.t_bits_off += [0] * (G - first)
        self.bits_ok += [False] * (G - first)
        if self.device.type != "cuda" or DENSE_MIN_FILL > 1.0:
            return
        nmax = int(lib.gnm_aggm_max_nodes())
        jobs = []                       # (graph, transposed?, rp_off, col_off, n, bits_off)
        pos = self.bits.size
        for g in range(first, G):
            n = self.n[g]
            if n < 1 or n > nmax:
                continue
            words = int(lib.gnm_adj_bits_words(n))
            jobs.append((g, False, self.rp_off[g], self.col_off[g], n, pos))
            self.bits_off[g] = pos
            pos += words
            if self.sym[g]:
                self.t_bits_off[g] = self.bits_off[g]
            else:
                jobs.append((g, True, self.t_rp_off[g], self.t_col_off[g], n, pos))
                self.t_bits_off[g] = pos
                pos += words
        if not jobs:
            return
        off = self.bits.reserve(pos - self.bits.size)
        assert off == jobs[0][5]
        d = self.device
        rp = torch.tensor([j[2] for j in jobs], dtype=torch.int64, device=d)
        co = torch.tensor([j[3] for j in jobs], dtype=torch.int64, device=d)
        nn = torch.tensor([j[4] for j in jobs], dtype=torch.int32, device=d)
        bo = torch.tensor([j[5] for j in jobs], dtype=torch.int64, device=d)
        dup = torch.empty(len(jobs), dtype=torch.int32, device=d)
        with torch.cuda.device(d):
            st = torch.cuda.current_stream(d).cuda_stream
            check(lib.gnm_adj_bits_build(self.rowptr.buf.data_ptr(), self.col.buf.data_ptr(), rp.data_ptr(),
                                         co.data_ptr(), nn.data_ptr(), len(jobs), self.bits.buf.data_ptr(),
                                         bo.data_ptr(), dup.data_ptr(), st), "gnm_adj_bits_build")
        dup_h = dup.cpu().numpy()
        bad = {jobs[k][0] for k in range(len(jobs)) if dup_h[k] != 0}
        for j in jobs:
            self.bits_ok[j[0]] = j[0] not in bad

    def add_many(self, graphs, threads=None):
        """add() for a whole dataset: host CSRs built on a few threads, one upload per array.  Returns the arena
        ids in order (graphs already in the arena keep theirs)."""
        import os
        from concurrent.futures import ThreadPoolExecutor
        ids = [None] * len(graphs)
        todo = []
        for i, g in enumerate(graphs):
            cached = getattr(g, "_gnm_cache", None)
            if cached is not None and cached[0] is self._token:
                ids[i] = cached[1]
            else:
                todo.append(i)

        def prep(i):
            g = graphs[i]
            n = len(g.g)
            em = g.edge_mat
            em = em.detach().cpu().numpy() if torch.is_tensor(em) else np.asarray(em)
            em = np.ascontiguousarray(em, dtype=np.int64).reshape(2, -1)
            feats = g.node_features
            feats = feats.detach().cpu() if torch.is_tensor(feats) else torch.as_tensor(np.asarray(feats))
            return n, self._host_csr(n, em), feats.to(torch.float32).contiguous()

        threads = threads or min(16, os.cpu_count() or 1)
        if len(todo) >= 32 and threads > 1:
            with ThreadPoolExecutor(max_workers=threads) as ex:
                items = list(ex.map(prep, todo))
        else:
            items = [prep(i) for i in todo]
        new = self._append_host(items)
        for i, gid in zip(todo, new):
            ids[i] = gid
            try:
                graphs[i]._gnm_cache = (self._token, gid)
            except Exception:
                pass
        return ids

    def _tables(self):
        if self._dev_tables is None:
            d = self.device
            t = lambda x, dt: torch.tensor(x, dtype=dt, device=d)
            self._dev_tables = dict(n=t(self.n, torch.int64), rp=t(self.rp_off, torch.int64),
                                    col=t(self.col_off, torch.int64), trp=t(self.t_rp_off, torch.int64),
                                    tcol=t(self.t_col_off, torch.int64), feat=t(self.feat_off, torch.int64),
                                    bits=t(self.bits_off, torch.int64), tbits=t(self.t_bits_off, torch.int64),
                                    bits_ok_host=np.asarray(self.bits_ok, dtype=bool),
                                    bits_host=np.asarray(self.bits_off, dtype=np.int64),
                                    tbits_host=np.asarray(self.t_bits_off, dtype=np.int64),
                                    n_host=np.asarray(self.n, dtype=np.int64),
                                    nnz_host=np.asarray(self.nnz, dtype=np.int64),
                                    sym_host=np.asarray(self.sym, dtype=bool),
                                    iso_host=np.asarray(self.iso, dtype=bool),
                                    rp_host=np.asarray(self.rp_off, dtype=np.int64),
                                    col_host=np.asarray(self.col_off, dtype=np.int64),
                                    trp_host=np.asarray(self.t_rp_off, dtype=np.int64),
                                    tcol_host=np.asarray(self.t_col_off, dtype=np.int64),
                                    feat_host=np.asarray(self.feat_off, dtype=np.int64))
        return self._dev_tables

    # ------------------------------------------------------------------ batches
    def batch(self, graphs):
        gids = self.add_many(graphs)
        return self.batch_from_gids(torch.tensor(gids, dtype=torch.int64))

    def batch_from_gids(self, gids):
        """gids: int64 tensor (host or device) of arena ids."""
        tb = self._tables()
        gh = gids.cpu().numpy() if torch.is_tensor(gids) else np.asarray(gids, dtype=np.int64)
        gd = self._upload(torch.as_tensor(gh, dtype=torch.int64))
        ns = tb["n_host"][gh]
        b = Batch()
        b.arena = self
        b.B = int(gh.shape[0])
        b.gids = gd
        b.n_max, b.n_min = int(ns.max()), int(ns.min())
        b.nnz_max = int(tb["nnz_host"][gh].max())
        node_off_host = np.zeros(b.B + 1, dtype=np.int64)
        np.cumsum(ns, out=node_off_host[1:])
        b.node_off_host = node_off_host
        b.N = int(node_off_host[-1])
        b.node_off = self._upload(torch.as_tensor(node_off_host.astype(np.int32)))
        b.rp_off = tb["rp"][gd]
        b.col_off = tb["col"][gd]
        b.feat_base = tb["feat"][gd]
        b.symmetric = bool(tb["sym_host"][gh].all())
        if b.symmetric:
            b.t_rp_off, b.t_col_off = b.rp_off, b.col_off
        else:
            b.t_rp_off, b.t_col_off = tb["trp"][gd], tb["tcol"][gd]
        b.dense = self.dense_ok(gh)
        b.iso = bool(tb["iso_host"][gh].any())
        b.has_bits = bool(gh.shape[0] > 0 and tb["bits_ok_host"][gh].all())
        b.bits_off = tb["bits"][gd]
        b.t_bits_off = b.bits_off if b.symmetric else tb["tbits"][gd]
        return b

    def _upload(self, host):
        """small host vector -> device without stalling the host: through pinned memory (torch's caching host allocator
        keeps the block until the copy has run).  A pageable source makes the copy wait for everything queued on the
        stream first -- with replayed steps in the queue that is a whole training step per batch assembled."""
        if self.device.type == "cuda":
            host = host.pin_memory()
        return host.to(self.device, non_blocking=True)

    def dense_ok(self, gh):
        """do the graphs gh (host int64 array of arena ids) form a batch for the matrix-core aggregation?"""
        tb = self._tables()
        if gh.shape[0] == 0 or not bool(tb["bits_ok_host"][gh].all()):
            return False
        ns = tb["n_host"][gh].astype(np.float64)
        return float(tb["nnz_host"][gh].sum()) >= DENSE_MIN_FILL * float((ns * ns).sum())

    def _feature_rows(self, batch):
        if batch.equal_n:
            base = batch.feat_base
            ar = self._arange.get(batch.n_max)
            if ar is None:
                ar = self._arange[batch.n_max] = torch.arange(batch.n_max, device=self.device)[None, :]
            return (base[:, None] + ar).reshape(-1)
        tb = self._tables()
        ns = tb["n"][batch.gids]
        base = torch.repeat_interleave(batch.feat_base, ns)
        start = torch.repeat_interleave(torch.as_tensor(batch.node_off_host[:-1], device=self.device), ns)
        return base + (torch.arange(batch.N, device=self.device) - start)

    def _gather_rows(self, batch, second=None):
        """rows of the batch's graphs out of the feature buffer (and out of `second`, an array of the same shape):
        one launch of gnm_gather_graph_rows (csrc/norm.hip) on a GPU arena; index_select on a host arena (tests)."""
        if self.device.type != "cuda":
            idx = self._feature_rows(batch)
            return self.feat.buf.index_select(0, idx), (second.index_select(0, idx) if second is not None else None)
        src = self.feat.buf
        W = src.shape[1]
        X = torch.empty((batch.N, W), dtype=torch.float32, device=self.device)
        Y = torch.empty_like(X) if second is not None else None
        check(lib.gnm_gather_graph_rows(src.data_ptr(), second.data_ptr() if second is not None else None, src.stride(0),
                                        W, batch.feat_base.data_ptr(), batch.node_off.data_ptr(), batch.B, X.data_ptr(),
                                        Y.data_ptr() if Y is not None else None, X.stride(0),
                                        torch.cuda.current_stream(self.device).cuda_stream), "gnm_gather_graph_rows")
        return X, Y

    def features(self, batch):
        """X_concat (graphcnn.py:195) gathered on the device: [N, F0] fp32."""
        return self._gather_rows(batch)[0]

    # layer 0's neighbour aggregation of the INPUT features does not depend on any parameter:
    #   learn_eps:  pooled_0 = A X [/deg] + (1 + eps_0) X      (graphcnn.py:154-161)  -> cache A X [/deg]
    #   otherwise:  pooled_0 = (A + I) X [/(deg + 1)]          (graphcnn.py:178-182)  -> cache all of it
    # so it is computed once per graph (by the same gnm_agg kernel, over the arena's own feature rows) and a
    # forward only gathers it, like the features themselves.
    AGG0_CACHE_BYTES = 8 << 30

    def _agg0_store(self, average, self_loop):
        if self.device.type != "cuda":
            raise GnmError("the GIN hot path runs on the GPU only (libgnm_hip.so); the arena is on %s" % self.device)
        key = (bool(average), bool(self_loop))
        store = self._agg0.setdefault(key, {"buf": None, "graphs": 0})
        G = len(self.n)
        if store["graphs"] == G:
            return store["buf"]
        rows_done = self.feat_off[store["graphs"]]          # rows of the graphs already cached
        if store["buf"] is None or store["buf"].shape[0] < self.feat.size:
            nb = torch.zeros((self.feat.buf.shape[0], self.feat.width), dtype=torch.float32, device=self.device)
            if store["buf"] is not None:
                nb[:rows_done].copy_(store["buf"][:rows_done])
            store["buf"] = nb
        if self._minus_one is None:
            self._minus_one = torch.full((1,), -1.0, dtype=torch.float32, device=self.device)
        minus_one = self._minus_one                                   # eps = -1: no self term
        st = torch.cuda.current_stream(self.device).cuda_stream
        F0 = self.feat.width
        for g0 in range(store["graphs"], G, 4096):
            g1 = min(G, g0 + 4096)
            bt = self.batch_from_gids(np.arange(g0, g1, dtype=np.int64))
            r0 = self.feat_off[g0]
            x = self.feat.buf[r0:r0 + bt.N]                 # feature rows are stored in arena (= this batch's) order
            y = store["buf"][r0:r0 + bt.N]
            # the same kernel choice as a step that aggregates the input features itself (gnm/core.py _agg): dense
            # batches take the matrix-core kernel, so cached and direct layer-0 values are the same bits
            rc = -2
            if bt.dense and (F0 < 32 or F0 % 32 == 0):
                rc = lib.gnm_aggm(self.rowptr.buf.data_ptr(), self.col.buf.data_ptr(), bt.rp_off.data_ptr(),
                                  bt.col_off.data_ptr(), self.bits.buf.data_ptr(), bt.bits_off.data_ptr(),
                                  self.rowptr.buf.data_ptr(), bt.rp_off.data_ptr(), bt.node_off.data_ptr(), bt.B,
                                  bt.n_max, x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), F0,
                                  None if self_loop else minus_one.data_ptr(), int(bool(average)),
                                  int(bool(self_loop)), 0, None, 0, None, st)
                if rc not in (0, -2):
                    check(rc, "gnm_aggm")
            if rc == -2:
                check(lib.gnm_agg(self.rowptr.buf.data_ptr(), self.col.buf.data_ptr(), bt.rp_off.data_ptr(),
                                  bt.col_off.data_ptr(), self.rowptr.buf.data_ptr(), bt.rp_off.data_ptr(),
                                  bt.node_off.data_ptr(), bt.B, bt.n_max, bt.nnz_max, x.data_ptr(), x.stride(0),
                                  y.data_ptr(), y.stride(0), F0, None if self_loop else minus_one.data_ptr(),
                                  int(bool(average)), int(bool(self_loop)), 0, None, 0, None, st), "gnm_agg")
        store["graphs"] = G
        return store["buf"]

    def refresh_agg0(self, average, self_loop):
        """Bring the layer-0 cache up to date with every graph in the arena, eagerly.  A replayed hipGraph only
        GATHERS from the cache: whoever replays one must call this first (it may re-allocate the cache, which the
        replayer's address check then notices)."""
        if self.feat is not None and self.feat.buf.numel() * 4 <= self.AGG0_CACHE_BYTES:
            self._agg0_store(average, self_loop)

    def features_and_agg0(self, batch, average, self_loop):
        """(X_concat, cached layer-0 aggregate) for the batch; the second is None when the cache would
        exceed AGG0_CACHE_BYTES (wide one-hot inputs on huge pools)."""
        if self.feat.buf.numel() * 4 > self.AGG0_CACHE_BYTES:
            return self._gather_rows(batch)[0], None
        store = self._agg0_store(average, self_loop)
        assert store.shape == self.feat.buf.shape and store.stride(0) == self.feat.buf.stride(0)
        return self._gather_rows(batch, store)

    # ------------------------------------------------------------------ parity export
    def export_adj_coo(self, batch, self_loops):
        """The reference's Adj_block._indices() (graphcnn.py:91-104) rebuilt from the
        arena, for the bit-exact index tests.  Host int64 [2, nnz(+N)]."""
        tb = self._tables()
        gh = batch.gids.cpu().numpy()
        nnz = int(np.asarray(self.nnz, dtype=np.int64)[gh].sum()) + (batch.N if self_loops else 0)
        rowptr_h = self.rowptr.buf[: self.rowptr.size].cpu().numpy()
        col_h = self.col.buf[: self.col.size].cpu().numpy().view(np.uint16)
        rp_off = batch.rp_off.cpu().numpy().astype(np.int64)
        col_off = batch.col_off.cpu().numpy().astype(np.int64)
        node_off = batch.node_off_host.astype(np.int32)
        out = np.empty((2, max(nnz, 1)), dtype=np.int64)
        k = lib.gnm_batch_coo_from_csr(rowptr_h.ctypes.data, col_h.ctypes.data, rp_off.ctypes.data,
                                       col_off.ctypes.data, node_off.ctypes.data, batch.B, int(bool(self_loops)),
                                       out[0].ctypes.data, out[1].ctypes.data)
        assert k == nnz, (k, nnz)
        return out[:, :nnz]


class BatchClassMismatch(ValueError):
    """StaticBatch.load: the batch is of another class (size, symmetry, density route, isolated nodes) than the one the
    buffers were captured for -- the one condition a replayed step answers by running eagerly (gnm/graphs.py)."""


class StaticBatch:
    """A Batch whose device tensors keep their addresses: load(batch) copies another
    batch's descriptors into them.  What a captured hipGraph of the training step reads
    (gnm/graphs.py); shapes (B, N, n_max) are fixed at construction.  The int64 vectors (and `extra`, e.g. the step's
    labels) are views of ONE buffer that load() fills with ONE concatenating launch: eight separate tiny copies,
    serialised in front of every replay, cost ~40 us of a 2.9 ms step."""

    def __init__(self, template, extra_int64=0):
        b = Batch()
        for f in ("B", "N", "n_max", "n_min", "nnz_max", "arena", "symmetric", "dense", "iso", "has_bits"):
            setattr(b, f, getattr(template, f))
        b.node_off_host = np.array(template.node_off_host, copy=True)
        b.node_off = template.node_off.clone()
        self._names = ["rp_off", "col_off", "gids", "feat_base", "bits_off"]
        if not template.symmetric:
            self._names += ["t_rp_off", "t_col_off", "t_bits_off"]
        B, k = template.B, len(self._names)
        self._packed = torch.empty(k * B + int(extra_int64), dtype=torch.int64, device=template.node_off.device)
        for j, name in enumerate(self._names):
            view = self._packed[j * B:(j + 1) * B]
            view.copy_(getattr(template, name))
            setattr(b, name, view)
        if template.symmetric:
            b.t_rp_off, b.t_col_off, b.t_bits_off = b.rp_off, b.col_off, b.bits_off
        self.extra = self._packed[k * B:] if extra_int64 else None
        if self.extra is not None:
            self.extra.zero_()
        self.batch = b

    def load(self, other, extra=None):
        b = self.batch
        if (other.B, other.N, other.n_max, other.n_min, other.symmetric, other.dense, other.iso) != \
                (b.B, b.N, b.n_max, b.n_min, b.symmetric, b.dense, b.iso) or other.nnz_max > b.nnz_max:
            raise BatchClassMismatch("StaticBatch.load: batch shape differs from the captured one")
        if (self.extra is None) != (extra is None):
            raise ValueError("StaticBatch.load: `extra` must be given exactly when the buffer was built with extra_int64")
        if not np.array_equal(other.node_off_host, b.node_off_host):       # equal-size graphs: never changes
            b.node_off.copy_(other.node_off, non_blocking=True)
            b.node_off_host = np.array(other.node_off_host, copy=True)
        if other is b:                      # its own descriptors: only the extra values change
            if extra is not None:
                self.extra.copy_(extra.reshape(-1), non_blocking=True)
            return
        srcs = [getattr(other, name) for name in self._names]
        if extra is not None:
            if extra.dtype != torch.int64 or extra.numel() != self.extra.numel():
                raise ValueError("StaticBatch.load: extra must be %d int64 values" % self.extra.numel())
            srcs.append(extra.reshape(-1))
        torch.cat(srcs, out=self._packed)


class PackedStaticBatch:
    """A fixed-shape Batch (B graphs of n nodes each) whose descriptors live in ONE device buffer that is refilled
    from the host with ONE pinned copy: load_gids(arena ids).  What the replayed evaluation forward reads
    (gnm/graphs.py CapturedEval): assembling a Batch the general way costs ~15 tiny device ops (~150 us of host
    time), which is most of a B = 1 forward."""

    def __init__(self, arena, B, n, symmetric, nnz_max, dense=False, iso=False, has_bits=False, extra_words=0,
                 two_stage=False):
        """extra_words: int64 words appended to the buffer for the caller's own per-batch values (a training step's
        labels and Infomax permutation: gnm/graphs.py), uploaded by the same copy: load_gids(gh, extra)."""
        dev = arena.device
        self.arena, self.B, self.n = arena, int(B), int(n)
        self._base_words = 8 * B + (B + 2) // 2           # 8 int64 vectors + node_off as int32 pairs
        words = self._base_words + int(extra_words)
        node_off = np.arange(B + 1, dtype=np.int64) * n
        # a small ring of pinned staging buffers: the copy is asynchronous, so a buffer may only be rewritten once
        # the copy that read it has run (an event per slot; by the time a slot comes round again it has)
        self._ring, self._events, self._next = [], [], 0
        for _ in range(32 if dev.type == "cuda" else 1):
            h = torch.zeros(words, dtype=torch.int64)
            if dev.type == "cuda":
                h = h.pin_memory()
            h.numpy()[8 * B:].view(np.int32)[:B + 1] = node_off       # constant: equal-size graphs
            self._ring.append(h)
            self._events.append(None)
        self._dev = torch.zeros(words, dtype=torch.int64, device=dev)
        # The upload itself runs on a side stream into a per-slot device staging buffer and reaches the buffer the
        # captured kernels read through a device-to-device copy on the caller's stream (round 4).  A host-to-device
        # copy queued on the compute stream is a DMA-engine transfer between two replays: it starts when the previous
        # replay has drained and the next one waits for it (~40 us per step measured); from the side stream it passes
        # while the previous step computes.  GNM_PACKED_DIRECT=1: the direct copy (A/B).
        # Only where a replay is LONG (a whole training step of hundreds of graphs: gnm/graphs.py CapturedTrainStep
        # asks for it): at one graph per forward the extra stream switch and event cost more host time than the
        # bubble they remove (0.130 -> 0.148 ms per evaluated graph measured with it everywhere).
        self._two_stage = bool(two_stage) and dev.type == "cuda" and os.environ.get("GNM_PACKED_DIRECT") is None
        self._side = torch.cuda.Stream(device=dev) if self._two_stage else None
        self._stage_dev = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in self._ring] if self._two_stage else None
        self._done = [None] * len(self._ring)
        dv = self._dev
        b = Batch()
        b.arena, b.B, b.N, b.n_max, b.n_min, b.nnz_max = arena, int(B), int(B * n), int(n), int(n), int(nnz_max)
        b.symmetric = bool(symmetric)
        b.dense = bool(dense)
        b.iso = bool(iso)
        b.has_bits = bool(has_bits)
        b.bits_off = dv[6 * B:7 * B]
        b.t_bits_off = b.bits_off if symmetric else dv[7 * B:8 * B]
        b.rp_off, b.col_off = dv[0:B], dv[B:2 * B]
        b.t_rp_off, b.t_col_off = (b.rp_off, b.col_off) if symmetric else (dv[2 * B:3 * B], dv[3 * B:4 * B])
        b.gids = dv[4 * B:5 * B]
        b.feat_base = dv[5 * B:6 * B]
        b.node_off = dv[8 * B:self._base_words].view(torch.int32)[:B + 1]
        b.node_off_host = node_off
        self.batch = b
        self.extra = dv[self._base_words:] if extra_words else None
        self._dev.copy_(self._ring[0])

    def fits(self, gh):
        """can the graphs with arena ids gh (host int64 array) be loaded? (same count, node count, symmetry class;
        no more edges than the launch parameters were sized for)"""
        tb = self.arena._tables()
        b = self.batch
        return (gh.shape[0] == b.B and bool((tb["n_host"][gh] == self.n).all())
                and bool(tb["sym_host"][gh].all()) == b.symmetric and int(tb["nnz_host"][gh].max()) <= b.nnz_max
                and self.arena.dense_ok(gh) == b.dense and bool(tb["iso_host"][gh].any()) == b.iso
                and bool(tb["bits_ok_host"][gh].all()) == b.has_bits)

    def load_gids(self, gh, extra=None):
        """extra: host int64 array filling the extra words (all of them), or None"""
        tb = self.arena._tables()
        i = self._next
        self._next = (i + 1) % len(self._ring)
        if self._events[i] is not None:
            self._events[i].synchronize()
        B = self.B
        hv = self._ring[i].numpy()
        np.take(tb["rp_host"], gh, out=hv[0:B])
        np.take(tb["col_host"], gh, out=hv[B:2 * B])
        np.take(tb["trp_host"], gh, out=hv[2 * B:3 * B])
        np.take(tb["tcol_host"], gh, out=hv[3 * B:4 * B])
        hv[4 * B:5 * B] = gh
        np.take(tb["feat_host"], gh, out=hv[5 * B:6 * B])
        np.take(tb["bits_host"], gh, out=hv[6 * B:7 * B])
        np.take(tb["tbits_host"], gh, out=hv[7 * B:8 * B])
        if extra is not None:
            hv[self._base_words:] = extra
        if self._two_stage:
            main = torch.cuda.current_stream(self._dev.device)
            if self._events[i] is None:
                self._events[i] = torch.cuda.Event()
            if self._done[i] is not None:
                self._side.wait_event(self._done[i])       # slot i's staging buffer: its last reader (on `main`) is through
            with torch.cuda.stream(self._side):
                self._stage_dev[i].copy_(self._ring[i], non_blocking=True)
                self._events[i].record(self._side)
            main.wait_event(self._events[i])
            self._dev.copy_(self._stage_dev[i], non_blocking=True)
            if self._done[i] is None:
                self._done[i] = torch.cuda.Event()
            self._done[i].record(main)
            return
        self._dev.copy_(self._ring[i], non_blocking=True)
        if self._dev.is_cuda:
            if self._events[i] is None:
                self._events[i] = torch.cuda.Event()
            self._events[i].record(torch.cuda.current_stream(self._dev.device))
