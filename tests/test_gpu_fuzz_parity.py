"""Randomised configurations of the whole hot path against the fp64 oracle ("parity unpinned" by reference goldens
for these shapes: the oracle itself is pinned by tests/test_oracle_golden.py).  The point is the DISPATCH matrix:
layer widths that do / do not take the 64-wide aggregation, its fused BatchNorm+ReLU+readout prologue and
BatchNorm-backward epilogue, the fused / narrow / generic Linear backward, 1-3 Linear MLPs, all pooling modes,
dropout masks, more than two classes -- every combination must agree with the same equations.

Tolerances: logits / loss 1e-5 relative (max-norm); gradients 5e-5 with the analytically-zero floor of helpers.py."""
import numpy as np
import pytest
import torch

from helpers import RTOL, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (seed, L, m, F0, H, C, B, n, density, learn_eps, gpool, npool)
CONFIGS = [
    (1, 3, 2, 7, 64, 2, 5, 48, 0.3, True, "sum", "sum"),            # agg16 + prologue + narrow fused Linear backward
    (2, 4, 2, 16, 64, 3, 3, 33, 0.5, True, "average", "average"),   # three classes, odd n
    (3, 2, 1, 5, 64, 2, 4, 40, 0.2, False, "sum", "average"),       # single-Linear MLPs, self loops
    (4, 3, 3, 9, 32, 2, 6, 21, 0.4, True, "sum", "sum"),            # H = 32: sliced aggregation kernel, 3-Linear MLPs
    (5, 5, 2, 64, 64, 2, 2, 64, 0.6, False, "average", "sum"),      # input as wide as the hidden layers
    (6, 1, 2, 3, 64, 4, 7, 17, 0.3, True, "sum", "average"),        # one layer only: no deferred readout at all
    (7, 3, 2, 40, 128, 2, 3, 30, 0.3, True, "sum", "sum"),          # H = 128: two-slice tiles, generic Linear backward
    (8, 2, 2, 7, 64, 2, 1, 400, 0.3, True, "sum", "sum"),           # one true-size graph
    (9, 3, 2, 7, 48, 2, 4, 25, 0.5, True, "average", "average"),    # H = 48: nothing 64-aligned
    # sparse graphs (below the fill at which batches go to the matrix cores) at H = 32: the CSR gather on 32-float slices
    # with the fused BatchNorm + ReLU + readout prologue and the fused backward statistics (round 4: gnm_agg_kernel<8, MODE>)
    (10, 3, 2, 7, 32, 2, 5, 60, 0.03, False, "average", "average"),
    (11, 4, 2, 7, 32, 3, 6, 70, 0.03, True, "average", "sum"),      # (learn_eps + neighbour "average" would put 0/0 rows
    (12, 3, 3, 7, 32, 2, 4, 64, 0.035, True, "sum", "sum"),         #  on the isolated nodes such graphs have)
]


class G:
    pass


def make_graphs(rng, B, n, F0, density):
    gs = []
    for _ in range(B):
        A = np.triu(rng.random((n, n)) < density, 1)
        iu, ju = np.nonzero(A)
        e = np.stack([iu, ju], 1)
        g = G()
        g.g = list(range(n))
        g.edge_mat = torch.from_numpy(np.ascontiguousarray(np.concatenate([e, e[:, ::-1]], 0).T.astype(np.int64)))
        g.node_features = torch.from_numpy(rng.standard_normal((n, F0)).astype(np.float32))
        g.label = 0
        gs.append(g)
    return gs


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "s%d_L%d_m%d_F%d_H%d_C%d" % c[:6])
def test_random_configuration_vs_fp64_oracle(cfg):
    from models.graphcnn import GIN_InfoMaxReg
    from oracle import gin_oracle as O
    seed, L, m, F0, H, C, B, n, density, learn_eps, gpool, npool = cfg
    rng = np.random.default_rng(seed)
    graphs = make_graphs(rng, B, n, F0, density)
    for g in graphs:
        g.label = int(rng.integers(0, C))
    dev = torch.device(DEV)
    torch.manual_seed(seed)
    p_drop = 0.4
    model = GIN_InfoMaxReg(L, m, F0, H, C, p_drop, learn_eps, gpool, npool, dev).to(dev).train()
    with torch.no_grad():
        model.eps.copy_(torch.from_numpy(rng.uniform(-0.3, 0.3, L).astype(np.float32)))
        for name, p in model.named_parameters():
            if "batch_norms" in name:
                p.add_(torch.from_numpy((0.1 * rng.standard_normal(tuple(p.shape))).astype(np.float32)).to(dev))
    state = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    perm = rng.permutation(B)
    # fixed dropout masks so that both sides apply the same ones (graphcnn.py:230): F.dropout is patched
    masks = ((rng.random((L, B, C)) >= p_drop) / (1 - p_drop)).astype(np.float32)
    import torch.nn.functional as F
    orig_dropout = F.dropout
    F.dropout = lambda x, p, training=True, inplace=False: torch.from_numpy(masks).to(x.device) * x
    try:
        bt = model.arena().batch(graphs)
        c_logit, d_logit = model.forward_batch(bt, perm=perm)
    finally:
        F.dropout = orig_dropout
    N = B * n
    labels = torch.tensor([g.label for g in graphs], device=dev)
    y = torch.cat([torch.ones(N, 1), torch.zeros(N, 1)]).to(dev)
    loss = torch.nn.functional.cross_entropy(c_logit, labels) + \
        0.05 * torch.nn.functional.binary_cross_entropy_with_logits(d_logit, y)
    loss.backward()
    torch.cuda.synchronize()

    om = O.OracleGIN(state, L, m, learn_eps, gpool, npool, dtype=np.float64)
    ob = [O.OGraph(n, g.edge_mat.numpy(), g.node_features.numpy(), g.label) for g in graphs]
    ref = om.train_step_grads(ob, perm, beta=0.05, dropout_masks=[masks[l].astype(np.float64) for l in range(L)])
    assert_close(c_logit.detach().cpu().numpy(), ref["c_logit"], rtol=RTOL, what="c_logit")
    assert_close(d_logit.detach().cpu().numpy(), ref["d_logit"], rtol=RTOL, what="d_logit")
    assert abs(loss.item() - ref["loss"]) <= RTOL * abs(ref["loss"])
    gmax = max(float(np.abs(v).max()) for k, v in ref["grads"].items() if not k.startswith("__"))
    n_checked = 0
    for name, p in model.named_parameters():
        if name == "eps" and not learn_eps:
            assert p.grad is None
            continue
        tg = np.asarray(ref["grads"][name]).reshape(tuple(p.shape))
        assert_close(p.grad.detach().cpu().numpy(), tg, rtol=5 * RTOL, what=name, floor=2e-2 * gmax)
        n_checked += 1
    assert n_checked >= 6
    for name, b in model.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(b) == 1
        else:
            assert_close(b.cpu().numpy(), om.p[name], rtol=RTOL, what=name)
