"""torch-CPU restatement of the GIN hot path -- TEST / BASELINE INFRASTRUCTURE ONLY.

The timed CPU baseline SURVEY.md 8(d) specifies: forward + backward on the host cores through the SAME ATen
operators the reference's Python calls (sparse-COO x dense `torch.spmm`, `F.linear`, `F.batch_norm`, `F.bilinear`,
autograd for the backward), with `torch.set_num_threads(os.cpu_count())`.  It is a functional restatement written
from the equations of SURVEY 8(a) -- no nn.Module tree, parameters come in as a plain `state_dict`-keyed dict of
arrays -- and, unlike oracle/gin_oracle.py (numpy, hand-derived backward, restructured discriminator), it keeps the
reference's COST STRUCTURE: an uncoalesced int64 COO adjacency rebuilt per forward, a degree spmm per layer, the
[N, LH] concatenation and `nn.Bilinear`'s trilinear autograd.  That makes it the like-for-like "reference CPU
GraphCNN.forward" number of the north_star; the numpy oracle stays the parity checker.

Only tests/ and bench.py's cpu_baseline leg may import it; the product package never does.

Parity status: PINNED by tests/test_torch_cpu_baseline.py against the golden vectors captured from the real
reference (tests/golden/make_goldens.py): logits / loss <= 1e-5 (max-norm), gradients 5e-5 on the tiny cases.
sum / average neighbour pooling only (the north_star path); "max" stays with gin_oracle.py.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _block_adjacency(batch, learn_eps):
    """graphcnn.py:84-106: uncoalesced block-diagonal COO, self loops appended when learn_eps is False."""
    start, idx = 0, []
    for g in batch:
        idx.append(torch.as_tensor(np.asarray(g.edge_mat), dtype=torch.int64) + start)      # :91
        start += int(g.num_nodes)
    idx = torch.cat(idx, 1)                                                                   # :92
    val = torch.ones(idx.shape[1])                                                            # :93
    if not learn_eps:                                                                         # :97-102
        loops = torch.arange(start, dtype=torch.int64)
        idx = torch.cat([idx, torch.stack([loops, loops])], 1)
        val = torch.cat([val, torch.ones(start)])
    return torch.sparse_coo_tensor(idx, val, (start, start), is_coalesced=False)             # :104


def _graph_pool(batch, mode):
    """graphcnn.py:109-134: [B, N] readout matrix, value 1 or 1/len(graph.g) (a Python float rounded to fp32)."""
    rows, cols, vals, start = [], [], [], 0
    for i, g in enumerate(batch):
        n = int(g.num_nodes)
        vals.extend([1.0 / n if mode == "average" else 1.0] * n)                              # :123 / :127
        rows.extend([i] * n)
        cols.extend(range(start, start + n))                                                  # :129
        start += n
    idx = torch.tensor([rows, cols], dtype=torch.int64)
    return torch.sparse_coo_tensor(idx, torch.tensor(vals, dtype=torch.float32), (len(batch), start))


class TorchCpuGIN:
    """GIN_InfoMaxReg.forward (graphcnn.py:194-251) + the reference's loss (main.py:34-37) on torch CPU tensors."""

    def __init__(self, state, num_layers, num_mlp_layers, learn_eps, graph_pooling_type, neighbor_pooling_type):
        assert neighbor_pooling_type in ("sum", "average")
        self.L, self.m, self.learn_eps = int(num_layers), int(num_mlp_layers), bool(learn_eps)
        self.gpool, self.npool = graph_pooling_type, neighbor_pooling_type
        self.p, self.buf = {}, {}
        for k, v in state.items():
            t = torch.as_tensor(np.asarray(v)).clone()
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                self.buf[k] = t
            else:
                self.p[k] = t.float().requires_grad_(True)

    def _bn(self, x, prefix, training):
        # nn.BatchNorm1d: batch statistics + running-stat update in train mode (mlp.py:48, graphcnn.py:163)
        if training:
            self.buf[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, self.buf[prefix + ".running_mean"], self.buf[prefix + ".running_var"],
                            self.p[prefix + ".weight"], self.p[prefix + ".bias"], training, 0.1, 1e-5)

    def _mlp(self, l, x, training):
        """mlp.py:40-49"""
        if self.m == 1:
            return F.linear(x, self.p[f"mlps.{l}.linear.weight"], self.p[f"mlps.{l}.linear.bias"])
        for k in range(self.m - 1):
            z = F.linear(x, self.p[f"mlps.{l}.linears.{k}.weight"], self.p[f"mlps.{l}.linears.{k}.bias"])
            x = F.relu(self._bn(z, f"mlps.{l}.batch_norms.{k}", training))
        k = self.m - 1
        return F.linear(x, self.p[f"mlps.{l}.linears.{k}.weight"], self.p[f"mlps.{l}.linears.{k}.bias"])

    def forward(self, batch, perm, training=True, want_disc=True):
        X = torch.cat([torch.as_tensor(np.asarray(g.node_features), dtype=torch.float32) for g in batch], 0)   # :195
        P = _graph_pool(batch, self.gpool)                                                    # :196
        n0 = int(batch[0].num_nodes)
        idx = [int(i) for i in perm for _ in range(n0)]                                       # :198-201 (graph ids!)
        A = _block_adjacency(batch, self.learn_eps)                                           # :206
        N, B = X.shape[0], len(batch)
        h, hidden = X, []
        for l in range(self.L):
            pooled = torch.spmm(A, h)                                                         # :154 / :178
            if self.npool == "average":
                pooled = pooled / torch.spmm(A, torch.ones(N, 1))                             # :155-158 / :181-182
            if self.learn_eps:
                pooled = pooled + (1 + self.p["eps"][l]) * h                                  # :161
            h = F.relu(self._bn(self._mlp(l, pooled, training), f"batch_norms.{l}", training))   # :162-166
            hidden.append(h)
        c_logit, latent = 0, []
        for l, hl in enumerate(hidden):                                                       # :228-231 (p = 0 dropout)
            ph = torch.spmm(P, hl)
            c_logit = c_logit + F.linear(ph, self.p[f"linears_prediction.{l}.weight"],
                                         self.p[f"linears_prediction.{l}.bias"])
            latent.append(ph)
        if not want_disc:
            return c_logit, None
        n_f, g_f = torch.cat(hidden, 1), torch.cat(latent, 1)                                 # :233-234
        c = torch.sigmoid(g_f)                                                                # :239
        shuf = n_f[idx, :]                                                                    # :241-242
        c_x = torch.cat([c[i].expand(N // B, -1) for i in range(B)], 0)                       # discriminator.py:23-26
        W, b = self.p["disc.f_k.weight"], self.p["disc.f_k.bias"]
        d_logit = torch.cat([F.bilinear(n_f, c_x, W, b), F.bilinear(shuf, c_x, W, b)], 0)     # :28-29, :36
        return c_logit, d_logit

    def train_step(self, batch, perm, beta=0.05, want_disc=True):
        """forward + CE + beta * BCE (main.py:34-37) + backward; returns loss, logits and the gradient dict."""
        for t in self.p.values():
            t.grad = None
        c_logit, d_logit = self.forward(batch, perm, True, want_disc)
        labels = torch.tensor([int(g.label) for g in batch], dtype=torch.int64)
        loss = F.cross_entropy(c_logit, labels)
        if want_disc:
            N = d_logit.shape[0] // 2
            d_labels = torch.cat([torch.ones(N, 1), torch.zeros(N, 1)], 0)
            loss = loss + beta * F.binary_cross_entropy_with_logits(d_logit, d_labels)
        loss.backward()
        return {"loss": float(loss.detach()), "c_logit": c_logit.detach().numpy(),
                "d_logit": None if d_logit is None else d_logit.detach().numpy(),
                "grads": {k: t.grad.numpy() for k, t in self.p.items() if t.grad is not None}}
