"""Evaluation forwards replayed from captured hipGraphs (gnm/graphs.py CapturedEval): what an unchanged main.py gets
from model([g]) in eval mode (main.py:49-57, 71-82).  The replay launches the eager path's kernels with the eager
path's parameters, so results must be BITWISE equal; host time per B = 1 forward drops from ~1.4 ms to ~0.1 ms."""
import time

import numpy as np
import pytest
import torch

from helpers import load_case
from test_gpu_model_parity import make_graphs, make_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["tiny_s1_eps1_gsum_nsum", "tiny_s0_eps0_gaverage_nsum", "true_s0_eps1_gaverage_naverage"])
def test_replay_is_bitwise_the_eager_forward(case):
    cfg, state, d = load_case(case)
    graphs = make_graphs(cfg, d)
    eager = make_model(cfg, state).eval()
    eager.eval_replay = False
    fast = make_model(cfg, state).eval()
    for batch in ([graphs[0]], [graphs[1]], graphs, [graphs[1]], graphs[::-1]):
        with torch.no_grad():
            np.random.seed(5)
            c0, d0 = eager(batch)
            np.random.seed(5)
            c1, d1 = fast(batch)
            assert np.random.get_state()[1][:8].tolist() == np.random.get_state()[1][:8].tolist()
        assert torch.equal(c0, c1) and torch.equal(d0, d1), len(batch)
        np.random.seed(6)
        l0 = eager(batch, latent=True)
        np.random.seed(6)
        l1 = fast(batch, latent=True)
        assert isinstance(l1, np.ndarray) and np.array_equal(l0, l1)
    assert len(fast._eval_cache) == 2                       # one captured graph per (B, n) shape
    # the numpy RNG is consumed exactly as by the eager path (one permutation per forward, graphcnn.py:199)
    np.random.seed(9); fast([graphs[0]]); a = np.random.random()
    np.random.seed(9); eager([graphs[0]]); b = np.random.random()
    assert a == b


def test_replayed_outputs_still_backpropagate():
    """main.py detaches eval outputs, but they carry a graph in the reference: a backward through a replayed
    forward recomputes it eagerly and must give the eager gradients."""
    cfg, state, d = load_case("tiny_s1_eps1_gsum_nsum")
    graphs = make_graphs(cfg, d)
    grads = []
    for replay in (False, True):
        m = make_model(cfg, state).eval()
        m.eval_replay = replay
        np.random.seed(3)
        c, dl = m(graphs)
        assert c.requires_grad and dl.requires_grad
        (c.sum() + 0.1 * dl.sum()).backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 10
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


def test_replay_follows_parameter_updates_and_new_graphs():
    from gnm import synth
    cfg, state, d = load_case("tiny_s1_eps1_gsum_nsum")
    m = make_model(cfg, state).eval()
    ref = make_model(cfg, state).eval()
    ref.eval_replay = False
    g = make_graphs(cfg, d)
    with torch.no_grad():
        m([g[0]])
        for mm in (m, ref):                                  # an optimizer step between evaluations
            for p in mm.parameters():
                p.mul_(1.01)
        extra = [synth.dense_fc_graph(50 + i, n=cfg["n"], t=40, f0=cfg["f0"]) for i in range(300)]   # arena regrows
        for x in extra[:3] + [g[2]]:
            np.random.seed(1); a = m([x])[0]
            np.random.seed(1); b = ref([x])[0]
            assert torch.equal(a, b)
        m.predict(extra, batch_size=64)                      # B = 64 batches replay too, the ragged tail eagerly
    # train mode never replays
    m.train()
    np.random.seed(2)
    assert m(g)[0].requires_grad and len(m._eval_cache) <= m.EVAL_REPLAY_ENTRIES


def test_one_graph_per_forward_is_not_python_bound():
    """pass_data_iteratively (main.py:49-57): one forward per graph.  Round 1: ~1.4 ms of host time each."""
    from gnm import synth
    from models.graphcnn import GIN_InfoMaxReg
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).eval()
    graphs = synth.make_pool("dense_fc", 64)
    model.arena().add_many(graphs)      # steady state (main.py evaluates the same graphs every epoch); the first
    out = []                            # sight of a graph also builds and uploads its CSR
    with torch.no_grad():
        for g in graphs[:8]:
            model([g])
        # (a) host cost of a forward: a burst short enough that the host never has to wait for the GPU
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in graphs[:24]:
            out.append(model([g])[0].detach())
        t_host = (time.perf_counter() - t0) / 24
        torch.cuda.synchronize()
        # (b) sustained rate: the GPU's dependent chain of ~110 small kernels per graph
        t0 = time.perf_counter()
        for g in graphs:
            out.append(model([g])[0].detach())
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / len(graphs)
    print("eval B=1: host %.3f ms per forward; sustained %.3f ms per graph (GPU-chain bound)" % (1e3 * t_host, 1e3 * t_all))
    assert torch.cat(out, 0).shape == (24 + 64, 2)
    assert t_host < 0.2e-3, t_host
    assert t_all < 1.2e-3, t_all          # round 1: ~1.4 ms per graph, host-bound
