#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REAL reference on CPU.

Run in the build container only (it needs /root/reference, which never travels
to the GPU box):

    python tests/golden/make_goldens.py

It imports /root/reference/models/graphcnn.py (GIN_InfoMaxReg, graphcnn.py:12),
feeds it seeded synthetic S2VGraph-shaped objects (field contract: util.py:9-17,
edge_mat built as util.py:99-103) and stores DATA ONLY -- inputs, the seeded
state_dict, and the reference's outputs -- as small .npz files next to this
script.  No reference source text is stored.

What is captured per case:
  * integer structures: Adj_block._indices() (graphcnn.py:84-106),
    graph_pool._indices()/values (graphcnn.py:109-134), the Infomax shuffle
    index list (graphcnn.py:198-201)                         -> bit-exact checks
  * per-layer `pooled` (input of mlps[l], graphcnn.py:161/178) and
    hidden_rep[l] (graphcnn.py:166/190), pooled_h, c_logit, d_logit, g_f
  * loss = CE + 0.05*BCE (main.py:34-37, d_labels sized by node count) and every
    parameter gradient after backward; BN running stats after the train step
  * compute_saliency output (graphcnn.py:254-299)
"""
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "models"))

import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
from models.graphcnn import GIN_InfoMaxReg  # noqa: E402  (the reference class)

HERE = os.path.dirname(os.path.abspath(__file__))
NP_SEED = 1234  # numpy global RNG seed set right before every reference forward


class SynthGraph:
    """Duck-typed stand-in for util.S2VGraph (util.py:9-17)."""

    def __init__(self, n, und_edges, feats, label):
        self.g = list(range(n))  # only len(graph.g) is read (graphcnn.py:90,116)
        self.label = int(label)
        self.node_features = torch.from_numpy(feats.astype(np.float32))
        e = np.asarray(und_edges, dtype=np.int64).reshape(-1, 2)
        both = np.concatenate([e, e[:, ::-1]], 0)  # util.py:99-100
        self.edge_mat = torch.from_numpy(np.ascontiguousarray(both.T))  # [2,E]
        self.neighbors = [[] for _ in range(n)]
        for i, j in e:
            self.neighbors[i].append(int(j))
            self.neighbors[j].append(int(i))
        self.max_neighbor = max(len(x) for x in self.neighbors)
        self.node_tags = None


def corr_graph(seed, n, t, f0, keep_pct=30.0):
    """SURVEY 8(d) generator: gaussian series -> corrcoef -> percentile mask
    over the whole matrix incl. diagonal (dataset.py:94), keep j>i (dataset.py:98)."""
    rng = np.random.default_rng(seed)
    ts = rng.standard_normal((t, n))
    cm = np.corrcoef(ts.T)
    mask = cm > np.percentile(cm, 100.0 - keep_pct)
    iu, ju = np.nonzero(np.triu(mask, 1))
    und = np.stack([iu, ju], 1)
    feats = rng.standard_normal((n, f0)).astype(np.float32)
    label = int(rng.integers(0, 2))
    return und, feats, label


def knn_graph(seed, n, k, f0):
    """SURVEY 8(d) sparse generator (BASELINE configs[3]): n points in 8 dimensions, each node's k nearest by squared
    L2 distance, symmetrised by union (reference graphs are always symmetric: util.py:99-100)."""
    rng = np.random.default_rng(seed)
    P = rng.standard_normal((n, 8))
    d2 = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    nn = np.argsort(d2, axis=1, kind="stable")[:, :k]
    A = np.zeros((n, n), dtype=bool)
    A[np.repeat(np.arange(n), k), nn.ravel()] = True
    A |= A.T
    iu, ju = np.nonzero(np.triu(A, 1))
    und = np.stack([iu, ju], 1)
    feats = rng.standard_normal((n, f0)).astype(np.float32)
    label = int(rng.integers(0, 2))
    return und, feats, label


def make_batch(base_seed, B, n, t, f0, isolate=None, knn=None, keep_pct=30.0):
    """isolate = (graph, node): that node loses all its edges (a row of padding only in the "max" pooling list);
    knn = k: k-nearest-neighbour graphs (knn_graph) instead of thresholded correlations"""
    graphs, raw = [], []
    for g in range(B):
        und, feats, label = (knn_graph(base_seed + g, n, knn, f0) if knn else
                             corr_graph(base_seed + g, n, t, f0, keep_pct))
        if isolate is not None and isolate[0] == g:
            und = und[(und != isolate[1]).all(1)]
        graphs.append(SynthGraph(n, und, feats, label))
        raw.append((und, feats, label))
    return graphs, raw


def build_model(seed, L, m, f0, H, C, dropout, learn_eps, gpool, npool):
    torch.manual_seed(seed)
    model = GIN_InfoMaxReg(L, m, f0, H, C, dropout, learn_eps, gpool, npool, torch.device("cpu"))
    # non-trivial eps / BN affine so their gradients and scaling are exercised;
    # drawn from a separate generator so the default init above is untouched.
    g = torch.Generator().manual_seed(seed + 77)
    with torch.no_grad():
        model.eps.copy_(0.1 * torch.randn(L, generator=g))
        for name, p in model.named_parameters():
            if "batch_norms" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            if "batch_norms" in name and name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return model


def run_case(tag, model_seed, graph_seed, B, n, t, L, m, f0, H, C, learn_eps, gpool, npool,
             state_file, row_stride=1, full_disc_grad=True, isolate=None, knn=None, no_isolated=False,
             keep_pct=30.0):
    if ONLY is not None and not tag.startswith(ONLY):
        return
    graphs, raw = make_batch(graph_seed, B, n, t, f0, isolate, knn, keep_pct)
    if no_isolated:      # neighbour "average" + learn_eps divides by the degree (graphcnn.py:157-158): keep 0/0 out
        for gr in graphs:
            assert min(len(x) for x in gr.neighbors) > 0, "%s: a node without neighbours; pick another graph seed" % tag
    out = {}
    out["cfg"] = np.array([L, m, f0, H, C, int(learn_eps), B, n], dtype=np.int64)
    out["gpool"] = np.array(gpool)
    out["npool"] = np.array(npool)
    out["np_seed"] = np.array(NP_SEED)
    out["row_stride"] = np.array(row_stride)
    for g, (und, feats, label) in enumerate(raw):
        out[f"und_{g}"] = und.astype(np.int16)
        out[f"feat_{g}"] = feats
    out["labels"] = np.array([r[2] for r in raw], dtype=np.int64)

    model = build_model(model_seed, L, m, f0, H, C, 0.0, learn_eps, gpool, npool)
    sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    if not os.path.exists(state_file):
        np.savez_compressed(state_file, **sd)
    else:  # the seeded init must not depend on the pooling / eps options
        old = np.load(state_file)
        for k in sd:
            assert np.array_equal(old[k], sd[k]), k
    out["state_file"] = np.array(os.path.basename(state_file))

    # ---- integer structures (private helpers, name-mangled) -----------------
    if npool != "max":
        adj = model._GIN_InfoMaxReg__preprocess_neighbors_sumavepool(graphs)
        out["adj_indices"] = adj._indices().numpy().astype(np.int64)
        out["adj_values"] = adj._values().numpy()
    else:
        # graphcnn.py:55-81 (graph.neighbors is rebuilt from und_* by the tests exactly as SynthGraph does above)
        out["padded_neighbors"] = model._GIN_InfoMaxReg__preprocess_neighbors_maxpool(graphs).numpy().astype(np.int64)
    gp = model._GIN_InfoMaxReg__preprocess_graphpool(graphs)
    out["gp_indices"] = gp._indices().numpy().astype(np.int64)
    out["gp_values"] = gp._values().numpy()

    np.random.seed(NP_SEED)
    perm = np.random.permutation(B)
    out["perm"] = perm.astype(np.int64)
    idx = []
    for i in perm:  # graphcnn.py:200-201
        idx += [i] * len(graphs[0].node_features)
    out["shuffle_idx"] = np.asarray(idx, dtype=np.int64)

    rs = slice(None, None, row_stride)

    def capture(model):
        store = {}
        hooks = []
        for l in range(L):
            hooks.append(model.mlps[l].register_forward_hook(
                lambda mod, inp, o, l=l: store.__setitem__(f"pooled_{l}", inp[0].detach().numpy().copy())))
            hooks.append(model.batch_norms[l].register_forward_hook(
                lambda mod, inp, o, l=l: store.__setitem__(f"hidden_{l}", torch.relu(o).detach().numpy().copy())))
        return store, hooks

    # ---- eval mode (fresh running stats) -------------------------------------
    model.eval()
    store, hooks = capture(model)
    np.random.seed(NP_SEED)
    with torch.no_grad():
        c_logit, d_logit = model(graphs)
    for h in hooks:
        h.remove()
    out["eval_c_logit"] = c_logit.numpy()
    out["eval_d_logit"] = d_logit.numpy()
    for l in range(L):
        out[f"eval_pooled_{l}"] = store[f"pooled_{l}"][rs]
        out[f"eval_hidden_{l}"] = store[f"hidden_{l}"][rs]
    np.random.seed(NP_SEED)
    out["eval_latent"] = model(graphs, latent=True)

    # saliency needs B == 1 (graphcnn.py:257)
    for cls in (0, 1):
        sal = model.compute_saliency([graphs[0]], cls)
        out[f"saliency_cls{cls}"] = sal.detach().numpy().copy()
    model.zero_grad()

    # ---- train mode, final_dropout = 0 ---------------------------------------
    model.train()
    store, hooks = capture(model)
    np.random.seed(NP_SEED)
    c_logit, d_logit = model(graphs)
    for h in hooks:
        h.remove()
    N = B * n
    c_labels = torch.from_numpy(out["labels"])
    d_labels = torch.cat([torch.ones(N, 1), torch.zeros(N, 1)], 0)  # main.py:32, sized by node count
    c_loss = torch.nn.CrossEntropyLoss()(c_logit, c_labels)
    d_loss = torch.nn.BCEWithLogitsLoss()(d_logit, d_labels)
    loss = c_loss + 0.05 * d_loss
    model.zero_grad()
    loss.backward()
    out["train_c_logit"] = c_logit.detach().numpy()
    out["train_d_logit"] = d_logit.detach().numpy()
    out["train_loss"] = np.array([loss.item(), c_loss.item(), d_loss.item()], dtype=np.float64)
    for l in range(L):
        out[f"train_pooled_{l}"] = store[f"pooled_{l}"][rs]
        out[f"train_hidden_{l}"] = store[f"hidden_{l}"][rs]
    rng = np.random.default_rng(99)
    for name, p in model.named_parameters():
        if p.grad is None:
            out[f"gradnone_{name}"] = np.array(1)
            continue
        g = p.grad.detach().numpy()
        if name == "disc.f_k.weight" and not full_disc_grad:
            W = g[0]
            r = rng.standard_normal(W.shape[1]).astype(np.float32)
            s = rng.standard_normal(W.shape[0]).astype(np.float32)
            out["gradproj_r"] = r
            out["gradproj_s"] = s
            out["gradproj_Wr"] = (W.astype(np.float64) @ r).astype(np.float64)
            out["gradproj_sW"] = (s @ W.astype(np.float64)).astype(np.float64)
            out["gradproj_fro"] = np.array(np.linalg.norm(W.astype(np.float64)))
        else:
            out[f"grad_{name}"] = g.copy()
    for name, b in model.named_buffers():
        out[f"bufafter_{name}"] = b.detach().numpy().copy()

    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"{tag}: {os.path.getsize(path)/1024:.0f} KB")


ONLY = None     # python tests/golden/make_goldens.py --only true_   regenerates the cases with that prefix


def main():
    global ONLY
    if "--only" in sys.argv:
        ONLY = sys.argv[sys.argv.index("--only") + 1]
    # seed 0: the reference's default dims (main.py:113-115) at F_0 = 7 (BASELINE configs)
    s0 = os.path.join(HERE, "state_seed0.npz")
    combos0 = [(True, "sum", "sum"), (True, "average", "average"),
               (False, "sum", "average"), (False, "average", "sum")]
    for le, gp, npl in combos0:
        run_case(f"tiny_s0_eps{int(le)}_g{gp}_n{npl}", 0, 500, B=3, n=24, t=48, L=5, m=2, f0=7, H=64, C=2,
                 learn_eps=le, gpool=gp, npool=npl, state_file=s0, full_disc_grad=False)
    # seed 1: small odd dims, every option combination
    s1 = os.path.join(HERE, "state_seed1.npz")
    for le in (True, False):
        for gp in ("sum", "average"):
            for npl in ("sum", "average"):
                run_case(f"tiny_s1_eps{int(le)}_g{gp}_n{npl}", 1, 700, B=4, n=20, t=40, L=3, m=2, f0=5, H=32, C=2,
                         learn_eps=le, gpool=gp, npool=npl, state_file=s1)
    # "max" neighbour pooling (graphcnn.py:55-81, 137-143; outside the north_star): both forms, one
    # batch with a node that has no neighbours (its pooled row is the dummy = column minimum, whose gradient goes to
    # torch.min's row).  ReLU zeros make equal maxima common from layer 1 on: the gradients pin the tie rule.
    run_case("tiny_s1_eps1_gsum_nmax", 1, 700, B=4, n=20, t=40, L=3, m=2, f0=5, H=32, C=2,
             learn_eps=True, gpool="sum", npool="max", state_file=s1)
    run_case("tiny_s1_eps0_gaverage_nmax", 1, 700, B=4, n=20, t=40, L=3, m=2, f0=5, H=32, C=2,
             learn_eps=False, gpool="average", npool="max", state_file=s1)
    run_case("tiny_s1_eps1_gaverage_nmax_iso", 1, 700, B=4, n=20, t=40, L=3, m=2, f0=5, H=32, C=2,
             learn_eps=True, gpool="average", npool="max", state_file=s1, isolate=(1, 3))
    run_case("tiny_s1_eps0_gsum_nmax_iso", 1, 700, B=4, n=20, t=40, L=3, m=2, f0=5, H=32, C=2,
             learn_eps=False, gpool="sum", npool="max", state_file=s1, isolate=(2, 0))
    # single-layer MLP variant (mlp.py:23-25, state_dict key mlps.l.linear.*)
    s2 = os.path.join(HERE, "state_seed2.npz")
    run_case("tiny_s2_mlp1", 2, 900, B=2, n=16, t=32, L=2, m=1, f0=4, H=32, C=2,
             learn_eps=True, gpool="sum", npool="sum", state_file=s2)
    # three-layer MLP variant (mlp.py:32-35)
    s3 = os.path.join(HERE, "state_seed3.npz")
    run_case("tiny_s3_mlp3", 3, 950, B=2, n=16, t=32, L=2, m=3, f0=6, H=32, C=2,
             learn_eps=False, gpool="average", npool="sum", state_file=s3)
    # true shape: n=400, 47,600 directed edges/graph; per-node arrays row-subsampled
    run_case("true_s0_eps1_gsum_nsum", 0, 1000, B=2, n=400, t=256, L=5, m=2, f0=7, H=64, C=2,
             learn_eps=True, gpool="sum", npool="sum", state_file=s0, row_stride=16, full_disc_grad=False)
    # the other aggregation / readout forms at true shape (round 2; BASELINE configs[4] mixes sum and average
    # pooling at n = 400): the self-loop form next_layer (graphcnn.py:170-191) with average neighbour pooling
    # (:181-182, divides by deg + 1) and sum readout; next_layer_eps with average neighbour pooling (:155-158)
    # and average readout (:122-123); the self-loop form with sum neighbours and average readout
    run_case("true_s0_eps0_gsum_naverage", 0, 1000, B=2, n=400, t=256, L=5, m=2, f0=7, H=64, C=2,
             learn_eps=False, gpool="sum", npool="average", state_file=s0, row_stride=16, full_disc_grad=False)
    run_case("true_s0_eps1_gaverage_naverage", 0, 1000, B=2, n=400, t=256, L=5, m=2, f0=7, H=64, C=2,
             learn_eps=True, gpool="average", npool="average", state_file=s0, row_stride=16, full_disc_grad=False)
    run_case("true_s0_eps0_gaverage_nsum", 0, 1000, B=2, n=400, t=256, L=5, m=2, f0=7, H=64, C=2,
             learn_eps=False, gpool="average", npool="sum", state_file=s0, row_stride=16, full_disc_grad=False)
    # round 4 -- more graphs than nodes per graph (B > n): the Infomax "shuffle" n_f[idx, :] (graphcnn.py:198-201,
    # 241-242) indexes ROWS 0 .. B-1 of the node matrix with graph numbers, so with B > n those rows leave graph 0 (the
    # regime BASELINE configs[1] runs: B = 1024 > n = 400; every case above has B < n).  B a multiple of n and not.
    run_case("tiny_s1_bgn_eps1_gsum_nsum", 1, 1300, B=12, n=8, t=16, L=3, m=2, f0=5, H=32, C=2,
             learn_eps=True, gpool="sum", npool="sum", state_file=s1)
    run_case("tiny_s1_bgn_eps0_gaverage_naverage", 1, 1300, B=16, n=8, t=16, L=3, m=2, f0=5, H=32, C=2,
             learn_eps=False, gpool="average", npool="average", state_file=s1)
    # ... at the headline widths (H = 64, five layers: the fused statistics kernels carry this branch), rows 0 .. 23
    # span graphs 0, 1 and part of 2
    run_case("tiny_s0_bgn_eps1_gsum_nsum", 0, 1400, B=24, n=10, t=24, L=5, m=2, f0=7, H=64, C=2,
             learn_eps=True, gpool="sum", npool="sum", state_file=s0, row_stride=3, full_disc_grad=False)
    run_case("tiny_s0_bgn_eps1_gaverage_naverage", 0, 1400, B=30, n=12, t=24, L=5, m=2, f0=7, H=64, C=2,
             learn_eps=True, gpool="average", npool="average", state_file=s0, row_stride=3, full_disc_grad=False,
             no_isolated=True, keep_pct=60.0)
    # round 4 -- BASELINE configs[3]'s shape: 1000-node kNN (k = 20, symmetrised by union) graphs, H = 128, five layers
    # (SURVEY 8(d) "Sparse synthetic"); per-node arrays row-subsampled.  Both aggregation forms: next_layer_eps with sum
    # pooling, and the self-loop form next_layer with average neighbour pooling and average readout.
    s4 = os.path.join(HERE, "state_seed4.npz")
    run_case("true_c4_s4_eps1_gsum_nsum", 4, 2000, B=2, n=1000, t=0, L=5, m=2, f0=7, H=128, C=2,
             learn_eps=True, gpool="sum", npool="sum", state_file=s4, row_stride=16, full_disc_grad=False, knn=20)
    run_case("true_c4_s4_eps0_gaverage_naverage", 4, 2000, B=2, n=1000, t=0, L=5, m=2, f0=7, H=128, C=2,
             learn_eps=False, gpool="average", npool="average", state_file=s4, row_stride=16, full_disc_grad=False,
             knn=20)


if __name__ == "__main__":
    main()
