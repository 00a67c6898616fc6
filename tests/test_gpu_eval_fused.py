"""The evaluation encoders -- one launch per graph (csrc/evalfwd.hip, gnm_eval_encoder; model.eval_fused = True) and one
launch per layer with a workgroup per 32-row block (csrc/evallayer.hip, gnm_eval_layers; model.eval_fused = "layers") --
under torch.no_grad(), what the reference's per-graph evaluation loop issues (main.py:49-57, 71-82): against the
layer-by-layer kernels, the fp64 oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from helpers import Calibrated, RTOL, assert_close, load_case
from test_gpu_kernels import RG, random_graphs
from test_gpu_model_parity import make_graphs, make_model, oracle_batch, oracle_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = [True, "layers"]
ENTRY = {True: "gnm_eval_encoder", "layers": "gnm_eval_layers"}


def random_model(L, m, f0, C, learn_eps, gpool, npool, seed, mode=True):
    from models.graphcnn import GIN_InfoMaxReg
    dev = torch.device(DEV)
    torch.manual_seed(seed)
    model = GIN_InfoMaxReg(L, m, f0, 64, C, 0.5, learn_eps, gpool, npool, dev).to(dev)
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():                  # non-trivial running statistics, BatchNorm affine parameters and eps
        for name, buf in model.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.3)
            elif name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
        for name, prm in model.named_parameters():
            if "batch_norms" in name:
                prm.copy_(torch.rand(prm.shape, generator=g) + 0.5 if name.endswith("weight")
                          else torch.randn(prm.shape, generator=g) * 0.2)
        model.eps.copy_(torch.randn(L, generator=g) * 0.3)
    model.eval_fused = mode                # (off by default: see models/graphcnn.py)
    return model.eval()


def count_calls(monkeypatch, name):
    from gnm import core
    calls = []
    real = getattr(core.lib, name)

    class Spy:
        def __call__(self, *a):
            calls.append(1)
            return real(*a)
    monkeypatch.setattr(core.lib, name, Spy(), raising=False)
    return calls


@pytest.mark.parametrize("L,m,f0,C,learn_eps,gpool,npool,sizes", [
    (5, 2, 7, 2, True, "sum", "sum", [400]),                   # the reference's evaluation call: one 400-node graph
    (5, 2, 7, 2, True, "sum", "sum", [400, 400, 400]),
    (3, 2, 7, 2, False, "average", "average", [400, 400]),
    (2, 1, 40, 3, True, "average", "sum", [57] * 4),           # one Linear per MLP, F0 spans two column blocks
    (3, 3, 64, 5, False, "sum", "average", [33] * 3),          # three Linears per MLP, F0 = 64
    (1, 2, 3, 2, True, "sum", "sum", [1, 1]),                   # single-node graphs
    (4, 2, 7, 2, True, "sum", "average", [200] * 66),          # more graphs than the replay cache takes: plain eager call
])
@pytest.mark.parametrize("mode", MODES)
def test_fused_eval_equals_the_layer_by_layer_path(L, m, f0, C, learn_eps, gpool, npool, sizes, mode, monkeypatch):
    rng = np.random.default_rng(L * 100 + m * 10 + f0)
    graphs = random_graphs(rng, sizes, 0.3, True, f0=f0)
    model = random_model(L, m, f0, C, learn_eps, gpool, npool, seed=L + m)
    model.eval_replay = False
    calls = count_calls(monkeypatch, ENTRY[mode])
    outs = []
    for fused in (mode, False):
        model.eval_fused = fused
        model._eval_cache = {}
        with torch.no_grad():
            np.random.seed(4)
            c_logit, d_logit = model(graphs)
            np.random.seed(4)
            lat = model(graphs, latent=True)
        outs.append((c_logit.cpu().numpy(), d_logit.cpu().numpy(), lat))
    assert len(calls) == 2                                     # (forward + latent) x the fused mode only
    # both are fp32 implementations of the same arithmetic whose rounding noise grows with depth (helpers.py): judge
    # each against the fp64 oracle -- the one-launch encoder may be no further from it than the layer-by-layer kernels
    # (x 3, floor 1e-5), and the two must agree with each other to that noise
    from oracle import gin_oracle as O
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    om = O.OracleGIN(state, L, m, learn_eps, gpool, npool, dtype=np.float64)
    ob = [O.OGraph(len(g.g), g.edge_mat.numpy(), g.node_features.numpy()) for g in graphs]
    np.random.seed(4)
    perm = np.random.permutation(len(graphs))
    with np.errstate(all="ignore"):
        tc, td, tcache = om.forward(ob, perm, training=False)
    truth = (tc, td, tcache["g_f"])
    from helpers import rel_err
    for a, b, t, what in zip(outs[0], outs[1], truth, ("c_logit", "d_logit", "latent")):
        assert a.shape == b.shape == t.shape
        e_fused, e_layer = rel_err(a, t), rel_err(b, t)
        assert e_fused <= max(1e-5, 3 * e_layer), "%s: fused %.2e vs layer-by-layer %.2e from the fp64 oracle" % (what, e_fused, e_layer)
        assert rel_err(a, b.astype(np.float64)) <= max(2e-5, 3 * (e_fused + e_layer)), what


@pytest.mark.parametrize("mode", MODES)
def test_fused_eval_vs_fp64_oracle_and_golden(mode):
    """true shape (B = 2, n = 400, H = 64, L = 5): the reference's own eval outputs and the fp64 oracle"""
    case = "true_s0_eps1_gsum_nsum"
    cfg, state, d = load_case(case)
    model = make_model(cfg, state).eval()
    model.eval_fused = mode
    graphs = make_graphs(cfg, d)
    with torch.no_grad():
        np.random.seed(cfg["np_seed"])
        c_logit, d_logit = model(graphs)
        np.random.seed(cfg["np_seed"])
        lat = model(graphs, latent=True)
    O, om = oracle_model(cfg, state)
    tc, td, tcache = om.forward(oracle_batch(O, cfg, d), d["perm"], training=False)
    cal = Calibrated()
    cal.check(lat, d["eval_latent"], tcache["g_f"], what="latent")
    cal.check(c_logit.cpu().numpy(), d["eval_c_logit"], tc, what="c_logit")
    cal.check(d_logit.cpu().numpy(), d["eval_d_logit"], td, what="d_logit")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", ["true_s0_eps0_gsum_naverage", "true_s0_eps1_gaverage_naverage", "true_s0_eps0_gaverage_nsum"])
def test_fused_eval_other_true_shape_goldens(case, mode):
    cfg, state, d = load_case(case)
    model = make_model(cfg, state).eval()
    model.eval_fused = mode
    graphs = make_graphs(cfg, d)
    with torch.no_grad():
        np.random.seed(cfg["np_seed"])
        c_logit, d_logit = model(graphs)
    O, om = oracle_model(cfg, state)
    tc, td, _ = om.forward(oracle_batch(O, cfg, d), d["perm"], training=False)
    cal = Calibrated()
    cal.check(c_logit.cpu().numpy(), d["eval_c_logit"], tc, what="c_logit")
    cal.check(d_logit.cpu().numpy(), d["eval_d_logit"], td, what="d_logit")


@pytest.mark.parametrize("mode", MODES)
def test_fused_eval_is_used_by_the_replayed_evaluation_and_repeatable(mode, monkeypatch):
    """the reference's loop: model([g]) per graph in eval mode (main.py:49-57) -- captured once, replayed; every
    replay launches the encoder, results are bitwise repeatable and equal the un-replayed call"""
    rng = np.random.default_rng(1)
    graphs = random_graphs(rng, [400] * 6, 0.3, True, f0=7)
    model = random_model(5, 2, 7, 2, True, "sum", "sum", seed=3, mode=mode)
    with torch.no_grad():
        first = [model([g])[0].clone() for g in graphs]
        again = [model([g])[0].clone() for g in graphs]
        model.eval_replay = False
        plain = [model([g])[0].clone() for g in graphs]
    for a, b, c in zip(first, again, plain):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_shapes_outside_the_encoder_fall_back(monkeypatch):
    """hidden 32, a graph of more than 400 nodes, a multigraph (no bit adjacency): the layer-by-layer kernels run"""
    from models.graphcnn import GIN_InfoMaxReg
    dev = torch.device(DEV)
    calls = count_calls(monkeypatch, "gnm_eval_encoder")
    rng = np.random.default_rng(2)
    torch.manual_seed(0)
    m32 = GIN_InfoMaxReg(2, 2, 5, 32, 2, 0.5, True, "sum", "sum", dev).to(dev).eval()
    m64 = GIN_InfoMaxReg(2, 2, 5, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).eval()
    m32.eval_fused = m64.eval_fused = True
    with torch.no_grad():
        m32(random_graphs(rng, [40, 40], 0.3, True, f0=5))
        m64(random_graphs(rng, [401, 401], 0.1, True, f0=5))
        em = np.array([[0, 1, 0, 1, 2], [1, 0, 1, 0, 0]])
        m64([RG(5, em, 5, rng), RG(5, em, 5, rng)])
    assert not calls
    with torch.no_grad():
        m64(random_graphs(rng, [40, 40], 0.3, True, f0=5))
    assert len(calls) >= 1             # (the replayed evaluation calls it in its warm-up passes and once under capture)


@pytest.mark.parametrize("H,sizes,f0", [(32, [100, 100], 5), (128, [65] * 3, 40), (64, [416, 416], 7), (64, [1], 2)])
def test_per_layer_eval_other_widths_and_sizes(H, sizes, f0):
    """gnm_eval_layers beyond the one-launch encoder's shapes: hidden 32 / 128, 416-node graphs (13 full row blocks), a
    single node -- against the layer-by-layer kernels"""
    from models.graphcnn import GIN_InfoMaxReg
    from helpers import rel_err
    dev = torch.device(DEV)
    rng = np.random.default_rng(H + len(sizes))
    torch.manual_seed(H)
    model = GIN_InfoMaxReg(3, 2, f0, H, 3, 0.5, True, "sum", "sum", dev).to(dev).eval()
    model.eval_replay = False
    graphs = random_graphs(rng, sizes, 0.2, True, f0=f0)
    outs = []
    for mode in ("layers", False):
        model.eval_fused = mode
        with torch.no_grad():
            np.random.seed(9)
            c_logit, d_logit = model(graphs)
            np.random.seed(9)
            lat = model(graphs, latent=True)
        outs.append((c_logit.cpu().numpy(), d_logit.cpu().numpy(), lat))
    for a, b, what in zip(outs[0], outs[1], ("c_logit", "d_logit", "latent")):
        assert a.shape == b.shape and np.isfinite(a).all()
        assert rel_err(a, b.astype(np.float64)) <= 3e-5, (what, rel_err(a, b.astype(np.float64)))


def test_backward_through_an_eval_mode_forward_on_the_encoder():
    """the reference's eval outputs carry an autograd graph (main.py:54 detaches them).  On the evaluation encoder the
    values come from csrc/evallayer.hip and the graph is attached lazily: a backward through them gives the gradients
    of the differentiable (layer-by-layer) eval forward."""
    rng = np.random.default_rng(11)
    graphs = random_graphs(rng, [60, 60, 60], 0.3, True, f0=7)
    grads = []
    for mode in ("layers", False):
        model = random_model(3, 2, 7, 2, True, "sum", "sum", seed=5, mode=mode)
        model.eval_replay = False
        model.zero_grad()
        np.random.seed(2)
        c_logit, d_logit = model(graphs)                       # grad mode, eval(): what main.py:49-57 does
        assert c_logit.requires_grad and d_logit.requires_grad
        (c_logit.square().sum() + 0.1 * d_logit.sum()).backward()
        grads.append({n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 20
    from helpers import rel_err
    for name in grads[0]:
        # (the upstream gradient 2 c_logit differs by the two forwards' rounding; everything else is the same backward)
        assert rel_err(grads[0][name], grads[1][name].astype(np.float64)) < 1e-4, name
