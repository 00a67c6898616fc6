"""Fused train-step tail through the C-ABI: gnm_loss_ce_bce and gnm_adam_step against the
oracle / torch's own ops, and the whole training trajectory against the golden trajectory captured from the
real reference driven by torch.optim.Adam + StepLR (tests/golden/train_s1.npz)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR, edge_mat_of, rel_err
from test_oracle_train_golden import noise_driven

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV)


@pytest.mark.parametrize("B,C,M,custom_target", [(4, 2, 160, False), (1, 2, 2, False), (37, 5, 4099, True),
                                                 (1024, 2, 819200, False), (3, 3, 0, False)])
def test_loss_kernel_vs_oracle_and_torch(B, C, M, custom_target):
    from gnm._cabi import check, lib
    from oracle.gin_oracle import OracleGIN
    rng = np.random.default_rng(B * 1000 + C)
    c_logit = (3 * rng.standard_normal((B, C))).astype(np.float32)
    d_logit = (4 * rng.standard_normal((M, 1))).astype(np.float32)
    if M:
        d_logit[0, 0] = 60.0            # saturated sigmoid on both sides
        d_logit[-1, 0] = -60.0
    labels = rng.integers(0, C, B)
    target = rng.random((M, 1)).astype(np.float32) if custom_target else \
        np.concatenate([np.ones((M // 2, 1)), np.zeros((M - M // 2, 1))]).astype(np.float32)
    beta = 0.05
    tc, td, tl = t(c_logit), t(d_logit), t(labels, torch.int64)
    tt = t(target) if custom_target else None
    loss3 = torch.empty(3, device=DEV)
    dC, dD = torch.empty_like(tc), torch.empty(M, device=DEV)
    ws = torch.empty(int(lib.gnm_loss_workspace_doubles(M)), dtype=torch.float64, device=DEV)
    check(lib.gnm_loss_ce_bce(tc.data_ptr(), tc.stride(0), tl.data_ptr(), B, C, td.data_ptr() if M else None,
                              tt.data_ptr() if tt is not None else None, M, M // 2, beta, loss3.data_ptr(),
                              dC.data_ptr(), dC.stride(0), dD.data_ptr() if M else None, ws.data_ptr(), None), "loss")
    torch.cuda.synchronize()
    got = loss3.cpu().numpy()
    if M:
        ref = OracleGIN.losses(c_logit.astype(np.float64), d_logit.astype(np.float64), labels,
                               target.astype(np.float64), beta)
        assert abs(got[0] - ref[0]) <= 2e-6 * abs(ref[0])
        assert abs(got[1] - ref[1]) <= 2e-6 * abs(ref[1])
        assert abs(got[2] - ref[2]) <= 2e-6 * abs(ref[2])
        assert rel_err(dC.cpu().numpy(), ref[3]) <= 2e-6
        assert rel_err(dD.cpu().numpy().reshape(M, 1), ref[4]) <= 2e-6
        # and torch's own ops on the device (what the reference's train() calls)
        tc2, td2 = tc.clone().requires_grad_(), td.clone().requires_grad_()
        lt = torch.nn.functional.cross_entropy(tc2, tl) + \
            beta * torch.nn.functional.binary_cross_entropy_with_logits(td2, t(target))
        lt.backward()
        assert abs(got[0] - lt.item()) <= 2e-6 * abs(lt.item())
        assert rel_err(dC.cpu().numpy(), tc2.grad.cpu().numpy()) <= 2e-6
        assert rel_err(dD.cpu().numpy(), td2.grad.view(-1).cpu().numpy()) <= 2e-6
    else:
        ce = torch.nn.functional.cross_entropy(tc, tl).item()
        assert abs(got[1] - ce) <= 2e-6 * abs(ce) and got[2] == 0.0 and abs(got[0] - got[1]) == 0.0


@pytest.mark.parametrize("scale", [1.0, -2.5])
def test_infomax_loss_autograd_matches_torch_losses_with_upstream_scale(scale):
    """infomax_loss(...).backward() with an upstream gradient != 1 (gnm_loss_ce_bce_grad reads it from the device)."""
    from gnm.train import infomax_loss
    rng = np.random.default_rng(7)
    B, C, N = 6, 2, 50
    c0, d0 = t(rng.standard_normal((B, C))), t(2 * rng.standard_normal((2 * N, 1)))
    lab = t(rng.integers(0, C, B), torch.int64)
    y = torch.cat([torch.ones(N, 1), torch.zeros(N, 1)]).to(DEV)
    grads = []
    for fused in (True, False):
        c, d = c0.clone().requires_grad_(), d0.clone().requires_grad_()
        if fused:
            loss, parts = infomax_loss(c, d, lab, 0.05)
            assert parts.requires_grad is False and abs(parts[0].item() - loss.item()) == 0.0
        else:
            loss = torch.nn.functional.cross_entropy(c, lab) + \
                0.05 * torch.nn.functional.binary_cross_entropy_with_logits(d, y)
        (loss * scale).backward()
        grads.append((loss.item(), c.grad.cpu().numpy(), d.grad.cpu().numpy()))
    assert abs(grads[0][0] - grads[1][0]) <= 2e-6 * abs(grads[1][0])
    assert rel_err(grads[0][1], grads[1][1]) <= 2e-6 and rel_err(grads[0][2], grads[1][2]) <= 2e-6


def test_loss_kernel_is_reproducible_bitwise():
    from gnm.train import infomax_loss
    rng = np.random.default_rng(0)
    c, d = t(rng.standard_normal((64, 2))), t(rng.standard_normal((2 * 64 * 400, 1)))
    lab = t(rng.integers(0, 2, 64), torch.int64)
    a = infomax_loss(c, d, lab)[1].clone()
    b = infomax_loss(c, d, lab)[1].clone()
    assert torch.equal(a, b)


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_kernel_vs_torch_adam(wd):
    from gnm._cabi import check, lib
    rng = np.random.default_rng(3)
    n = 142_288 + 3                       # the model's parameter count at H = 64, plus a ragged tail
    p0 = rng.standard_normal(n).astype(np.float32)
    grads = [(rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 1, n)).astype(np.float32) for _ in range(5)]
    lrs = [0.01, 0.01, 0.005, 0.005, 0.0025]
    ref_p = t(p0).clone().requires_grad_()
    opt = torch.optim.Adam([ref_p], lr=lrs[0], weight_decay=wd)
    p, m, v = t(p0).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    for g, lr in zip(grads, lrs):
        opt.param_groups[0]["lr"] = lr
        ref_p.grad = t(g)
        opt.step()
        hyper = torch.tensor([lr, 0.9, 0.999, 1e-8, wd, 1.0], dtype=torch.float64, device=DEV)
        tg = t(g)
        check(lib.gnm_adam_step(p.data_ptr(), tg.data_ptr(), m.data_ptr(), v.data_ptr(), n, hyper.data_ptr(),
                                step.data_ptr(), None), "adam")
        torch.cuda.synchronize()
    assert int(step.item()) == 5
    st = opt.state[ref_p]
    assert rel_err(m.cpu().numpy(), st["exp_avg"].cpu().numpy()) <= 1e-6
    assert rel_err(v.cpu().numpy(), st["exp_avg_sq"].cpu().numpy()) <= 1e-6
    # the parameters themselves: at most one fp32 ulp of the parameter apart (division vs reciprocal-multiply
    # inside the update), and the moves agree to 1e-4 of the largest move
    pn, rn = p.cpu().numpy(), ref_p.detach().cpu().numpy()
    assert np.all(np.abs(pn - rn) <= 1.2e-7 * np.maximum(np.abs(rn), 1.0))
    assert np.max(np.abs((pn - p0) - (rn - p0))) <= 1e-4 * np.max(np.abs(rn - p0))


def test_adam_grad_scale_equals_prescaled_gradient():
    from gnm._cabi import check, lib
    rng = np.random.default_rng(5)
    n = 1000
    res = []
    for scale, mult in ((0.25, 1.0), (1.0, 0.25)):        # 0.25 is exact in binary: both orders round alike
        p, m, v = t(rng.standard_normal(n) * 0 + 1), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        g = t(np.linspace(-3, 3, n)) * mult
        step = torch.zeros(1, dtype=torch.int32, device=DEV)
        hyper = torch.tensor([1e-2, 0.9, 0.999, 1e-8, 0.0, scale], dtype=torch.float64, device=DEV)
        check(lib.gnm_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, hyper.data_ptr(),
                                step.data_ptr(), None), "adam")
        res.append(p.clone())
    assert torch.equal(res[0], res[1])


class G:
    pass


def load_train_case():
    d = dict(np.load(os.path.join(GOLDEN_DIR, "train_s1.npz")))
    L, m, f0, H, C, le, B, n = [int(x) for x in d["cfg"]]
    state = dict(np.load(os.path.join(GOLDEN_DIR, "state_seed1.npz")))
    graphs = []
    for g in range(B):
        o = G()
        o.g = list(range(n))
        o.edge_mat = torch.from_numpy(edge_mat_of(d[f"und_{g}"]))
        o.node_features = torch.from_numpy(d[f"feat_{g}"])
        o.label = int(d["labels"][g])
        graphs.append(o)
    return d, dict(L=L, m=m, f0=f0, H=H, C=C, B=B, n=n), state, graphs


def fresh_model(cfg, state):
    from models.graphcnn import GIN_InfoMaxReg
    dev = torch.device(DEV)
    m = GIN_InfoMaxReg(cfg["L"], cfg["m"], cfg["f0"], cfg["H"], cfg["C"], 0.0, True, "sum", "sum", dev)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    return m.to(dev).train()


@pytest.mark.parametrize("capture", [False, True])
def test_training_trajectory_vs_reference_golden(capture):
    """forward + fused loss + backward + fused Adam + StepLR for 6 steps == the reference model under
    torch.optim.Adam / StepLR (golden), step by step; eager launches and hipGraph replay."""
    from gnm.train import FusedTrainStep, StepLR
    d, cfg, state, graphs = load_train_case()
    lr, beta, step_size, gamma = [float(x) for x in d["hyper"]]
    model = fresh_model(cfg, state)
    arena = model.arena()
    gids = np.array([arena.add(g) for g in graphs], dtype=np.int64)
    batch = arena.batch_from_gids(gids)
    labels = torch.tensor([g.label for g in graphs], device=DEV)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    step = FusedTrainStep(model, lr=lr, beta=beta, template_batch=batch, capture=capture)
    for k, v in model.state_dict().items():                     # construction (warm-up + capture) has no side effect
        assert torch.equal(v, before[k]), k
    assert int(step.optimizer.step_count.item()) == 0 and float(step.optimizer.exp_avg.abs().max()) == 0.0
    sched = StepLR(step.optimizer, int(step_size), gamma)
    ref = d["losses"]
    for s in range(int(d["steps"])):
        assert abs(step.optimizer.lr - d["lrs"][s]) < 1e-15
        parts = step.run(batch, labels, d["perms"][s]).cpu().numpy()
        sched.step()
        for j in range(3):
            assert abs(parts[j] - ref[s, j]) <= 5e-4 * max(abs(ref[s, j]), 1e-2), (s, j, parts, ref[s])
    assert int(step.optimizer.step_count.item()) == int(d["steps"])
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        if k.endswith("running_mean") or noise_driven(k, cfg["m"]):
            continue                                              # see tests/test_oracle_train_golden.py
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(d["final_" + k])
        else:
            assert rel_err(v, d["final_" + k]) <= 5e-3, k


def test_captured_trajectory_equals_eager_bitwise():
    from gnm.train import FusedTrainStep
    d, cfg, state, graphs = load_train_case()
    out = []
    for capture in (False, True):
        model = fresh_model(cfg, state)
        arena = model.arena()
        batch = arena.batch_from_gids(np.array([arena.add(g) for g in graphs], dtype=np.int64))
        labels = torch.tensor([g.label for g in graphs], device=DEV)
        step = FusedTrainStep(model, lr=0.01, template_batch=batch, capture=capture)
        parts = [step.run(batch, labels, d["perms"][s]).clone() for s in range(4)]
        out.append((torch.stack(parts), step.dp.fp.flat.clone(), step.optimizer.exp_avg_sq.clone()))
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)


def test_fused_loss_gives_the_same_gradients_as_torch_losses():
    from gnm.parallel import DataParallelGIN
    from gnm.train import infomax_loss
    d, cfg, state, graphs = load_train_case()
    flats = []
    for fused in (False, True):
        model = fresh_model(cfg, state)
        dp = DataParallelGIN(model)
        np.random.seed(3)
        c, dl = model(graphs)
        lab = torch.tensor([g.label for g in graphs], device=DEV)
        if fused:
            loss, parts = infomax_loss(c, dl, lab, 0.05)
        else:
            N = dl.shape[0] // 2
            y = torch.cat([torch.ones(N, 1), torch.zeros(N, 1)]).to(DEV)
            loss = torch.nn.functional.cross_entropy(c, lab) + \
                0.05 * torch.nn.functional.binary_cross_entropy_with_logits(dl, y)
        loss.backward()
        flats.append((loss.item(), dp.fp.flat_grad.clone().cpu().numpy()))
    assert abs(flats[0][0] - flats[1][0]) <= 2e-6 * abs(flats[0][0])
    assert rel_err(flats[1][1], flats[0][1]) <= 5e-6


@pytest.mark.parametrize("case,scale", [("tiny_s1_eps1_gsum_nsum", 1.0), ("tiny_s1_eps1_gaverage_naverage", -0.7),
                                        ("true_s0_eps1_gsum_nsum", 1.0)])
def test_fused_loss_takes_the_discriminator_hand_over(case, scale, monkeypatch):
    """With gnm.train.infomax_loss (default targets) the model's backward scales the reductions its score kernel left
    in the forward (gnm/core.py DiscUnit) instead of running gnm_disc_score_bwd's pass over the hidden layers.  Every
    gradient must equal the route through that pass (the same loss with the hand-over switched off, and torch's own
    losses as main.py:34-37 writes them) to fp32 rounding, and the pass must really be skipped / really run."""
    from gnm import core
    from gnm._cabi import lib
    from gnm.train import infomax_loss
    from helpers import load_case
    from test_gpu_model_parity import make_graphs, make_model
    cfg, state, d = load_case(case)
    graphs = make_graphs(cfg, d)
    labels = torch.tensor([g.label for g in graphs], device=DEV)
    calls = []
    real = lib.gnm_disc_score_bwd

    class Spy:
        def __call__(self, *a):
            calls.append(1)
            return real(*a)
    monkeypatch.setattr(core.lib, "gnm_disc_score_bwd", Spy(), raising=False)

    def run(mode):
        model = make_model(cfg, state).train()
        model.train_replay = False          # the hand-over lives on the eager path (a replayed backward is captured once,
        np.random.seed(3)                   # before any loss exists, with the general pass in it)
        c_logit, d_logit = model(graphs)
        if mode == "torch":
            N = d_logit.shape[0] // 2
            y = torch.cat([torch.ones(N, 1), torch.zeros(N, 1)]).to(DEV)
            loss = torch.nn.functional.cross_entropy(c_logit, labels) + \
                0.05 * torch.nn.functional.binary_cross_entropy_with_logits(d_logit, y)
        else:
            if mode == "nohold":
                del d_logit._gnm_disc_unit
            loss = infomax_loss(c_logit, d_logit, labels, 0.05)[0]
        n0 = len(calls)
        (loss * scale).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters() if p.grad is not None}, len(calls) - n0

    g_unit, n_unit = run("unit")
    g_pass, n_pass = run("nohold")
    g_torch, n_torch = run("torch")
    assert (n_unit, n_pass, n_torch) == (0, 1, 1)
    gmax = max(float(np.abs(v).max()) for v in g_pass.values())
    for name in g_pass:
        assert rel_err(g_unit[name], g_pass[name], floor=2e-2 * gmax) <= 2e-5, name
        assert rel_err(g_unit[name], g_torch[name], floor=2e-2 * gmax) <= 2e-5, name


def test_discriminator_hand_over_is_not_used_when_the_gradient_was_summed():
    """d_logit feeding a second consumer: autograd sums two gradients into a new tensor, which is not k (sigmoid - t):
    the backward must notice (pointer check) and run the general pass."""
    from gnm.train import infomax_loss
    from helpers import load_case
    from test_gpu_model_parity import make_graphs, make_model
    cfg, state, d = load_case("tiny_s1_eps1_gsum_nsum")
    graphs = make_graphs(cfg, d)
    labels = torch.tensor([g.label for g in graphs], device=DEV)
    res = []
    for hand_over in (True, False):
        model = make_model(cfg, state).train()
        model.train_replay = False
        np.random.seed(3)
        c_logit, d_logit = model(graphs)
        if not hand_over:
            del d_logit._gnm_disc_unit
        loss = infomax_loss(c_logit, d_logit, labels, 0.05)[0] + 1e-3 * d_logit.square().sum()
        loss.backward()
        res.append({k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()})
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k          # the same kernels ran: bitwise


def test_discriminator_hand_over_is_not_used_when_a_hook_edits_the_gradient_in_place():
    """an in-place tensor hook on d_logit (g.mul_(0.5)) keeps the gradient's address, so the pointer check
    alone took the k x unit shortcut and ignored the edit for dU / dWd / the readout path.  The gradient's version
    counter is part of the hand-over now: same bits as the run without any hand-over."""
    from gnm.train import infomax_loss
    from helpers import load_case
    from test_gpu_model_parity import make_graphs, make_model
    cfg, state, d = load_case("tiny_s1_eps1_gsum_nsum")
    graphs = make_graphs(cfg, d)
    labels = torch.tensor([g.label for g in graphs], device=DEV)
    res = []
    for hand_over in (True, False):
        model = make_model(cfg, state).train()
        model.train_replay = False
        np.random.seed(3)
        c_logit, d_logit = model(graphs)
        if not hand_over:
            del d_logit._gnm_disc_unit
        d_logit.register_hook(lambda g: g.mul_(0.5))
        infomax_loss(c_logit, d_logit, labels, 0.05)[0].backward()
        res.append({k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()})
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
