"""CPU oracle for the GIN message-passing hot path -- TEST INFRASTRUCTURE ONLY.

This is a numpy/scipy restatement (no torch, no autograd) of the algorithm in
the reference's models/graphcnn.py, models/mlp.py and models/discriminator.py,
with a hand-derived backward.  It exists to CHECK the HIP path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the
product package (graph-neural-mapping_amd/) never does.

Parity status: PINNED.  tests/test_oracle_golden.py compares every function
below against golden vectors captured from the real reference run on CPU
(tests/golden/make_goldens.py): integer structures bit-exact, floats within
1e-5 relative (max-norm), for float32 and float64 arithmetic.

The arithmetic the reference delegates to PyTorch ATen (third-party, not under
/root/reference; README.md:26 pins only "pytorch >= 1.4.0", the goldens were
made with torch 2.10.0) is restated from its documented semantics:
torch.spmm on an uncoalesced COO matrix (duplicates add), nn.Linear (x W^T + b),
nn.BatchNorm1d (biased batch variance to normalise, unbiased to update running
stats, eps 1e-5, momentum 0.1), nn.Bilinear (x1^T W x2 + b), F.dropout.

Each function cites the reference file:line it follows.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default (constructed at mlp.py:38, graphcnn.py:51)
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- #
# integer structures ("pooling indices": must be bit-exact)                    #
# --------------------------------------------------------------------------- #
def start_indices(batch):
    """graphcnn.py:88-90 / :112-116 -- cumulative node offsets, len(graph.g) each."""
    start = [0]
    for g in batch:
        start.append(start[-1] + int(g.num_nodes))
    return np.asarray(start, dtype=np.int64)


def build_adj_block(batch, learn_eps):
    """graphcnn.py:84-106 -- block-diagonal COO index/value arrays (uncoalesced,
    edge_mat order, self loops appended when learn_eps is False)."""
    start = start_indices(batch)
    idx = np.concatenate([np.asarray(g.edge_mat, dtype=np.int64) + start[i]
                          for i, g in enumerate(batch)], axis=1)            # :91-92
    val = np.ones(idx.shape[1], dtype=np.float32)                            # :93
    if not learn_eps:                                                        # :97-102
        n = int(start[-1])
        loops = np.stack([np.arange(n, dtype=np.int64)] * 2)
        idx = np.concatenate([idx, loops], axis=1)
        val = np.concatenate([val, np.ones(n, dtype=np.float32)])
    return idx, val, (int(start[-1]), int(start[-1]))


def build_graph_pool(batch, graph_pooling_type):
    """graphcnn.py:109-134 -- [B, N] readout COO; value 1 or 1/len(graph.g)."""
    start = start_indices(batch)
    rows, cols, vals = [], [], []
    for i, g in enumerate(batch):
        n = int(g.num_nodes)
        if graph_pooling_type == "average":
            vals.extend([1.0 / n] * n)                                       # :123 (python float -> fp32)
        else:
            vals.extend([1] * n)                                             # :127
        rows.extend([i] * n)
        cols.extend(range(int(start[i]), int(start[i + 1])))                 # :129
    idx = np.stack([np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)])
    return idx, np.asarray(vals, dtype=np.float32), (len(batch), int(start[-1]))


def build_padded_neighbors(batch, learn_eps):
    """graphcnn.py:55-81 -- [N, max_deg (+1)] padded neighbour lists: batch-global ids in graph.neighbors order,
    -1 in the padding slots (:69), the node itself appended when learn_eps is False (:73-74)."""
    max_deg = max(int(g.max_neighbor) for g in batch)                         # :59
    start = start_indices(batch)
    rows = []
    for i, g in enumerate(batch):
        for j in range(len(g.neighbors)):
            pad = [int(x) + int(start[i]) for x in g.neighbors[j]]            # :67
            pad.extend([-1] * (max_deg - len(pad)))                          # :69
            if not learn_eps:
                pad.append(j + int(start[i]))                                 # :74
            rows.append(pad)
    return np.asarray(rows, dtype=np.int64).reshape(len(rows), max_deg + (0 if learn_eps else 1))


def maxpool_fwd(h, padded):
    """graphcnn.py:137-143 -- max over the padded neighbour rows, the padding pointing at a dummy row that holds the
    column minimum of h.  Returns (pooled, cache); the cache records WHICH candidate torch.max / torch.min select
    (ATen's CPU kernels keep the first of several equal extrema -- as np.argmax / np.argmin do, NaN included),
    because that is where autograd sends the gradient."""
    if padded.shape[1] == 0:
        raise IndexError("max(): Expected reduction dim 1 to have non-zero size.")    # what torch raises at :142
    N = h.shape[0]
    amin = np.argmin(h, axis=0)                                               # :140 torch.min(h, dim = 0)
    dummy = h[amin, np.arange(h.shape[1])]
    hd = np.concatenate([h, dummy[None, :]], 0)                               # :141
    cand = hd[padded]                                                         # :142  [N, D, F] (-1 -> the dummy row)
    sel = np.argmax(cand, axis=1)                                             # [N, F] position within the padded row
    pooled = np.take_along_axis(cand, sel[:, None, :], 1)[:, 0, :]
    src = np.take_along_axis(padded, sel, 1)                                  # selected row id, -1 = dummy
    return pooled, dict(src=src, amin=amin, N=N)


def maxpool_bwd(dpooled, cache):
    """autograd of maxpool_fwd: each element's gradient goes to the selected row; the dummy's share continues to
    the row torch.min selected for that column."""
    N, F_ = cache["N"], dpooled.shape[1]
    dhd = np.zeros((N + 1, F_), dtype=dpooled.dtype)
    cols = np.broadcast_to(np.arange(F_), cache["src"].shape)
    np.add.at(dhd, (np.where(cache["src"] < 0, N, cache["src"]), cols), dpooled)
    dh = dhd[:N]
    np.add.at(dh, (cache["amin"], np.arange(F_)), dhd[N])
    return dh


def shuffle_index(batch, perm):
    """graphcnn.py:198-201 -- graph index repeated len(batch[0].node_features)
    times.  NOTE: these are GRAPH indices later used as ROW indices of n_f
    (graphcnn.py:242); the quirk is reproduced, not fixed."""
    n0 = int(batch[0].node_features.shape[0])
    return np.repeat(np.asarray(perm, dtype=np.int64), n0)


# --------------------------------------------------------------------------- #
# ATen op restatements                                                         #
# --------------------------------------------------------------------------- #
def coo_to_csr(idx, val, shape, dtype):
    """torch.spmm semantics on an uncoalesced COO: duplicate entries add."""
    return sp.coo_matrix((val.astype(dtype), (idx[0], idx[1])), shape=shape).tocsr()


def linear_fwd(x, W, b):
    """nn.Linear: y = x W^T + b (mlp.py:25,32-35; graphcnn.py:52)."""
    return x @ W.T + b


def bn_fwd(x, gamma, beta, run_mean, run_var, training):
    """nn.BatchNorm1d forward.  Returns y, cache, (new_run_mean, new_run_var)."""
    N = x.shape[0]
    if training:
        mean = x.mean(axis=0)
        var = ((x - mean) ** 2).mean(axis=0)                # biased, used to normalise
        unbiased = var * (N / max(N - 1, 1))
        new_rm = (1 - BN_MOMENTUM) * run_mean + BN_MOMENTUM * mean
        new_rv = (1 - BN_MOMENTUM) * run_var + BN_MOMENTUM * unbiased
    else:
        mean, var = run_mean.astype(x.dtype), run_var.astype(x.dtype)
        new_rm, new_rv = run_mean, run_var
    rstd = 1.0 / np.sqrt(var + x.dtype.type(BN_EPS))
    xhat = (x - mean) * rstd
    y = xhat * gamma + beta
    return y, (xhat, rstd, gamma, training), (new_rm, new_rv)


def bn_bwd(dy, cache):
    xhat, rstd, gamma, training = cache
    dgamma = (dy * xhat).sum(axis=0)
    dbeta = dy.sum(axis=0)
    if training:
        N = dy.shape[0]
        dx = gamma * rstd * (dy - dbeta / N - xhat * (dgamma / N))
    else:
        dx = dy * (gamma * rstd)
    return dx, dgamma, dbeta


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


class OGraph:
    """Field contract of util.S2VGraph the hot path reads (util.py:9-17)."""

    def __init__(self, n, edge_mat, node_features, label=0, neighbors=None, max_neighbor=None):
        self.num_nodes = int(n)                        # len(graph.g)
        self.edge_mat = np.asarray(edge_mat, dtype=np.int64)
        self.node_features = np.asarray(node_features, dtype=np.float32)
        self.label = int(label)
        # read by the "max" pooling path only (graphcnn.py:59-67)
        self.neighbors = neighbors
        self.max_neighbor = (max((len(x) for x in neighbors), default=0) if max_neighbor is None and neighbors is not None
                             else max_neighbor)


# --------------------------------------------------------------------------- #
# the model                                                                    #
# --------------------------------------------------------------------------- #
class OracleGIN:
    """Restates GIN_InfoMaxReg (graphcnn.py:12-299): sum / average / max neighbour pooling."""

    def __init__(self, state, num_layers, num_mlp_layers, learn_eps,
                 graph_pooling_type, neighbor_pooling_type, dtype=np.float32):
        assert neighbor_pooling_type in ("sum", "average", "max")
        self.L, self.m = int(num_layers), int(num_mlp_layers)
        self.learn_eps = bool(learn_eps)
        self.gpool, self.npool = graph_pooling_type, neighbor_pooling_type
        self.dtype = np.dtype(dtype)
        self.p = {k: (np.asarray(v).astype(dtype) if np.asarray(v).dtype.kind == "f" else np.asarray(v).copy())
                  for k, v in state.items()}

    # -- parameter access (state_dict keys) ----------------------
    def _lin(self, l, k):
        if self.m == 1:
            return self.p[f"mlps.{l}.linear.weight"], self.p[f"mlps.{l}.linear.bias"]
        return self.p[f"mlps.{l}.linears.{k}.weight"], self.p[f"mlps.{l}.linears.{k}.bias"]

    def _bn(self, prefix):
        return (self.p[prefix + ".weight"], self.p[prefix + ".bias"],
                self.p[prefix + ".running_mean"], self.p[prefix + ".running_var"])

    def _bn_apply(self, x, prefix, training, update):
        g, b, rm, rv = self._bn(prefix)
        y, cache, (nrm, nrv) = bn_fwd(x, g, b, rm, rv, training)
        if training and update:
            self.p[prefix + ".running_mean"] = nrm.astype(self.dtype)
            self.p[prefix + ".running_var"] = nrv.astype(self.dtype)
            self.p[prefix + ".num_batches_tracked"] = self.p[prefix + ".num_batches_tracked"] + 1
        return y, cache

    # -- forward ---------------------------------------------------------------
    def forward(self, batch, perm, training=False, dropout_masks=None, update_running=True, want_disc=True):
        """graphcnn.py:194-251.  `perm` is the np.random.permutation(B) the
        reference draws at :199.  dropout_masks: optional list of [B,C] keep/(1-p)
        multipliers (F.dropout at :230); None means p == 0 or eval mode.
        want_disc=False stops after the classifier (:231) and returns d_logit = None: the
        "encoder + classifier only" variant of the timed CPU baseline."""
        dt = self.dtype
        L, m = self.L, self.m
        B = len(batch)
        X = np.concatenate([g.node_features for g in batch], 0).astype(dt)   # :195
        N = X.shape[0]
        gp_idx, gp_val, gp_shape = build_graph_pool(batch, self.gpool)       # :196
        # the reference holds 1/n rounded to fp32 (:130); gp_val keeps that rounding in any dtype
        P = coo_to_csr(gp_idx, gp_val, gp_shape, dt)
        idx = shuffle_index(batch, perm)                                      # :198-201
        A = deg = padded = None
        if self.npool == "max":
            padded = build_padded_neighbors(batch, self.learn_eps)            # :204
        else:
            a_idx, a_val, a_shape = build_adj_block(batch, self.learn_eps)    # :206
            A = coo_to_csr(a_idx, a_val, a_shape, dt)
            deg = np.asarray(A.sum(axis=1)).reshape(-1, 1).astype(dt)         # :157/:181 spmm(A, ones)

        cache = dict(A=A, P=P, deg=deg, idx=idx, B=B, N=N, layers=[], X=X, training=training)
        h = X
        hidden = []
        for l in range(L):
            lc = {"h_in": h}
            if self.npool == "max":
                pooled, lc["max"] = maxpool_fwd(h, padded)                    # :151 / :175
            else:
                pooled = A @ h                                                # :154 / :178
            if self.npool == "average":
                with np.errstate(divide="ignore", invalid="ignore"):
                    pooled = pooled / deg                                     # :158 / :182 (0/0 -> NaN kept)
            if self.learn_eps:
                pooled = pooled + (1 + self.p["eps"][l]) * h                  # :161
            lc["pooled"] = pooled
            x = pooled
            lc["mlp"] = []
            if m == 1:                                                        # mlp.py:41-43
                W, b = self._lin(l, 0)
                lc["mlp"].append(("lin", x))
                x = linear_fwd(x, W, b)
            else:
                for k in range(m - 1):                                        # mlp.py:47-48
                    W, b = self._lin(l, k)
                    z = linear_fwd(x, W, b)
                    y, bnc = self._bn_apply(z, f"mlps.{l}.batch_norms.{k}", training, update_running)
                    lc["mlp"].append(("lin_bn_relu", x, bnc, y > 0))
                    x = np.maximum(y, 0)
                W, b = self._lin(l, m - 1)                                    # mlp.py:49
                lc["mlp"].append(("lin", x))
                x = linear_fwd(x, W, b)
            y, bnc = self._bn_apply(x, f"batch_norms.{l}", training, update_running)   # :163 / :187
            lc["bn_out"] = (bnc, y > 0)
            h = np.maximum(y, 0)                                              # :166 / :190
            lc["h_out"] = h
            hidden.append(h)
            cache["layers"].append(lc)

        c_logit = 0
        pooled_h = []
        for l, hl in enumerate(hidden):                                       # :228-231
            ph = P @ hl
            lg = linear_fwd(ph, self.p[f"linears_prediction.{l}.weight"], self.p[f"linears_prediction.{l}.bias"])
            if training and dropout_masks is not None:
                lg = lg * dropout_masks[l]
            c_logit = c_logit + lg
            pooled_h.append(ph)
        if not want_disc:
            cache.update(hidden=hidden, pooled_h=pooled_h, n_f=None, g_f=np.concatenate(pooled_h, 1),
                         dropout_masks=dropout_masks if training else None)
            return c_logit, None, cache
        n_f = np.concatenate(hidden, 1)                                       # :233
        g_f = np.concatenate(pooled_h, 1)                                     # :234
        c = sigmoid(g_f)                                                      # :239
        shuf = n_f[idx, :]                                                    # :241-242
        # Discriminator.forward, discriminator.py:19-38
        rep = N // B                                                          # :24 integer division
        c_x = np.repeat(c, rep, axis=0)                                       # :23-26
        if c_x.shape[0] != N:
            raise ValueError("Discriminator needs equal-size graphs (discriminator.py:24)")
        Wd = self.p["disc.f_k.weight"][0]
        bd = self.p["disc.f_k.bias"][0]
        t1 = n_f @ Wd
        t2 = shuf @ Wd
        sc1 = (t1 * c_x).sum(1, keepdims=True) + bd                           # :28  nn.Bilinear
        sc2 = (t2 * c_x).sum(1, keepdims=True) + bd                           # :29
        d_logit = np.concatenate([sc1, sc2], 0)                               # :36
        cache.update(hidden=hidden, pooled_h=pooled_h, n_f=n_f, g_f=g_f, c=c, shuf=shuf, c_x=c_x,
                     t1=t1, t2=t2, dropout_masks=dropout_masks if training else None)
        return c_logit, d_logit, cache

    # -- backward (what torch autograd derives from main.py:40) ----------------
    def backward(self, cache, d_c_logit, d_d_logit, want_dx=False):
        dt = self.dtype
        L, m = self.L, self.m
        B, N = cache["B"], cache["N"]
        H = cache["hidden"][0].shape[1]
        A, P, deg = cache["A"], cache["P"], cache["deg"]
        grads = {}
        dC = np.asarray(d_c_logit, dtype=dt)
        with_disc = cache["n_f"] is not None and d_d_logit is not None
        dn_f = dg_f = None
        if with_disc:
            dD = np.asarray(d_d_logit, dtype=dt)
            dsc1, dsc2 = dD[:N], dD[N:]
            Wd = self.p["disc.f_k.weight"][0]
            n_f, shuf, c_x, c = cache["n_f"], cache["shuf"], cache["c_x"], cache["c"]
            # nn.Bilinear backward: dW = x1^T (dy * x2); dx1 = dy * (x2 W^T); dx2 = dy * (x1 W)
            grads["disc.f_k.weight"] = (n_f.T @ (dsc1 * c_x) + shuf.T @ (dsc2 * c_x))[None]
            grads["disc.f_k.bias"] = np.array([dsc1.sum() + dsc2.sum()], dtype=dt)
            cW = c_x @ Wd.T
            dn_f = dsc1 * cW
            np.add.at(dn_f, cache["idx"], dsc2 * cW)                          # gather backward (graphcnn.py:242)
            dc_x = dsc1 * cache["t1"] + dsc2 * cache["t2"]
            dc = dc_x.reshape(B, N // B, -1).sum(1)
            dg_f = dc * c * (1 - c)                                           # sigmoid backward
        dh_next = None
        for l in reversed(range(L)):
            lc = cache["layers"][l]
            Wp = self.p[f"linears_prediction.{l}.weight"]
            dlg = dC if cache["dropout_masks"] is None else dC * cache["dropout_masks"][l]
            grads[f"linears_prediction.{l}.weight"] = dlg.T @ cache["pooled_h"][l]
            grads[f"linears_prediction.{l}.bias"] = dlg.sum(0)
            dph = dlg @ Wp
            if with_disc:
                dph = dph + dg_f[:, l * H:(l + 1) * H]
            dh = P.T @ dph
            if with_disc:
                dh = dh + dn_f[:, l * H:(l + 1) * H]
            if dh_next is not None:
                dh = dh + dh_next
            bnc, mask = lc["bn_out"]
            dy = dh * mask
            dx, dg, db = bn_bwd(dy, bnc)
            grads[f"batch_norms.{l}.weight"], grads[f"batch_norms.{l}.bias"] = dg, db
            for k in reversed(range(len(lc["mlp"]))):
                ent = lc["mlp"][k]
                W, _ = self._lin(l, k)
                wname = f"mlps.{l}.linear" if m == 1 else f"mlps.{l}.linears.{k}"
                if ent[0] == "lin":
                    xin = ent[1]
                    grads[wname + ".weight"] = dx.T @ xin
                    grads[wname + ".bias"] = dx.sum(0)
                    dx = dx @ W
                else:
                    _, xin, bnc_k, mask_k = ent
                    # here dx is the grad wrt relu(bn(lin(xin)))
                    dyk = dx * mask_k
                    dz, dgk, dbk = bn_bwd(dyk, bnc_k)
                    grads[f"mlps.{l}.batch_norms.{k}.weight"] = dgk
                    grads[f"mlps.{l}.batch_norms.{k}.bias"] = dbk
                    grads[wname + ".weight"] = dz.T @ xin
                    grads[wname + ".bias"] = dz.sum(0)
                    dx = dz @ W
            dpooled = dx
            h_in = lc["h_in"]
            d_h = np.zeros_like(h_in)
            if self.learn_eps:
                grads.setdefault("eps", np.zeros(L, dtype=dt))
                # heavily cancelling reduction (terms >> result): accumulate in fp64 whatever
                # the working dtype, as the HIP kernel does
                grads["eps"][l] = (dpooled.astype(np.float64) * h_in).sum()
                d_h = d_h + (1 + self.p["eps"][l]) * dpooled
            if self.npool == "max":
                d_h = d_h + maxpool_bwd(dpooled, lc["max"])
            else:
                src = dpooled / deg if self.npool == "average" else dpooled
                d_h = d_h + A.T @ src
            dh_next = d_h
        if want_dx:
            grads["__dX"] = dh_next
        return grads

    # -- losses (main.py:16-17, 34-37) ------------------------------------------
    @staticmethod
    def losses(c_logit, d_logit, c_labels, d_labels, beta):
        """CrossEntropyLoss(mean) + beta * BCEWithLogitsLoss(mean); returns loss and
        the gradients wrt both logits."""
        z = c_logit - c_logit.max(1, keepdims=True)
        lse = np.log(np.exp(z).sum(1, keepdims=True))
        logp = z - lse
        Bn = c_logit.shape[0]
        c_loss = -logp[np.arange(Bn), c_labels].mean()
        dC = np.exp(logp)
        dC[np.arange(Bn), c_labels] -= 1
        dC /= Bn
        x, y = d_logit, d_labels
        d_loss = (np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))).mean()
        dD = (sigmoid(x) - y) / x.size
        return c_loss + beta * d_loss, c_loss, d_loss, dC, beta * dD

    def train_step_grads(self, batch, perm, beta=0.05, dropout_masks=None, update_running=True, want_disc=True):
        """One main.py:29-40 step without the optimizer: forward (train mode),
        loss, backward.  d_labels sized by node count (SURVEY 8(b) caller quirk).
        want_disc=False: the loss is CrossEntropy(c_logit) alone and the Infomax tail is not computed."""
        c_logit, d_logit, cache = self.forward(batch, perm, training=True, dropout_masks=dropout_masks,
                                               update_running=update_running, want_disc=want_disc)
        N = cache["N"]
        c_labels = np.asarray([g.label for g in batch], dtype=np.int64)
        if not want_disc:
            loss, c_loss, _, dC, _ = self.losses(c_logit, np.zeros((1, 1), self.dtype), c_labels,
                                                 np.zeros((1, 1), self.dtype), 0.0)
            grads = self.backward(cache, dC, None)
            return dict(loss=c_loss, c_loss=c_loss, d_loss=0.0, c_logit=c_logit, d_logit=None, grads=grads, cache=cache)
        d_labels = np.concatenate([np.ones((N, 1)), np.zeros((N, 1))], 0).astype(self.dtype)
        loss, c_loss, d_loss, dC, dD = self.losses(c_logit, d_logit, c_labels, d_labels, beta)
        grads = self.backward(cache, dC, dD)
        return dict(loss=loss, c_loss=c_loss, d_loss=d_loss, c_logit=c_logit, d_logit=d_logit,
                    grads=grads, cache=cache)

    # -- optimizer (main.py:136-137, 39-41, 153) ------------------------------------
    def param_names(self):
        """trainable tensors, i.e. everything in the state dict that is not a BatchNorm buffer"""
        return [k for k in self.p if not (k.endswith("running_mean") or k.endswith("running_var")
                                          or k.endswith("num_batches_tracked"))]

    def adam_step(self, grads, opt_state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """optimizer.step() of optim.Adam(model.parameters(), lr) (main.py:136, 41).  torch.optim.Adam
        (third party) restated from its documented update: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
        p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps).  opt_state: {'t': int, name: (m, v)}."""
        t = opt_state["t"] = opt_state.get("t", 0) + 1
        b1, b2 = betas
        for k in self.param_names():
            if k not in grads:
                continue
            g = np.asarray(grads[k], dtype=self.dtype).reshape(self.p[k].shape)
            if weight_decay:
                g = g + weight_decay * self.p[k]
            m, v = opt_state.get(k, (np.zeros_like(self.p[k]), np.zeros_like(self.p[k])))
            m = (m + (g - m) * self.dtype.type(1 - b1)).astype(self.dtype)
            v = (v * self.dtype.type(b2) + self.dtype.type(1 - b2) * g * g).astype(self.dtype)
            denom = np.sqrt(v) / self.dtype.type(np.sqrt(1 - b2 ** t)) + self.dtype.type(eps)
            self.p[k] = (self.p[k] - self.dtype.type(lr / (1 - b1 ** t)) * (m / denom)).astype(self.dtype)
            opt_state[k] = (m, v)
        return opt_state

    def train_trajectory(self, batch, perms, lr, beta=0.05, step_size=None, gamma=1.0):
        """len(perms) steps of the reference's train() body (main.py:29-41) on one fixed batch, with
        StepLR(step_size, gamma) stepped after every optimizer step.  Returns [[loss, c_loss, d_loss], ...]."""
        opt_state, losses = {}, []
        for s, perm in enumerate(perms):
            r = self.train_step_grads(batch, perm, beta=beta)
            losses.append([float(r["loss"]), float(r["c_loss"]), float(r["d_loss"])])
            cur = lr * (gamma ** (s // step_size)) if step_size else lr
            self.adam_step(r["grads"], opt_state, cur)
        return np.array(losses), opt_state

    def latent(self, batch, perm):
        """forward(..., latent=True), graphcnn.py:248-249."""
        _, _, cache = self.forward(batch, perm, training=False)
        return cache["g_f"]

    def compute_saliency(self, graph, cls):
        """graphcnn.py:254-299: eval mode, B == 1, grad of the class score wrt X."""
        c_logit, d_logit, cache = self.forward([graph], np.array([0]), training=False)
        dC = np.zeros_like(c_logit)
        dC[0, cls] = 1                                                        # :263-264
        # the saliency graph does not include the discriminator (:287-296)
        grads = self.backward(cache, dC, np.zeros_like(d_logit), want_dx=True)
        return grads["__dX"]
