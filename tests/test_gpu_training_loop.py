"""End-to-end use of the drop-in module the way the reference's main.py drives it
(restated here; main.py itself needs tensorboard and the private HCP data): Adam + StepLR over
GIN_InfoMaxReg.parameters() (main.py:136-137), train() steps with CE + beta*BCE and the per-step
.cpu() sync (main.py:19-47), per-graph evaluation in eval mode (main.py:49-57,85-96), latent export
(main.py:71-82), saliency (main.py:60-68), state_dict save/load (main.py:166)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_task(num, n=48, f0=7, seed=0):
    """two classes of correlation graphs whose node features differ in mean: learnable in a few steps"""
    from gnm import synth
    rng = np.random.default_rng(seed)
    graphs = []
    for g in range(num):
        gr = synth.dense_fc_graph(10_000 + g, n=n, t=64, f0=f0)
        gr.label = g % 2
        gr.node_features = gr.node_features + (0.8 if gr.label else -0.8)
        graphs.append(gr)
    return graphs


def test_training_loop_learns_and_eval_paths_work(tmp_path):
    from models.graphcnn import GIN_InfoMaxReg
    dev = torch.device(DEV)
    torch.manual_seed(0)
    np.random.seed(0)
    train_graphs, test_graphs = make_task(64), make_task(16, seed=1)
    model = GIN_InfoMaxReg(3, 2, 7, 32, 2, 0.5, True, "sum", "sum", dev).to(dev)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.005)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=5, gamma=0.8)
    c_criterion, d_criterion = torch.nn.CrossEntropyLoss(), torch.nn.BCEWithLogitsLoss()
    batch_size, beta = 16, 0.05
    n = len(train_graphs[0].g)
    losses = []
    for epoch in range(6):
        model.train()
        for _ in range(8):
            sel = np.random.permutation(len(train_graphs))[:batch_size]
            batch = [train_graphs[i] for i in sel]
            c_logit, d_logit = model(batch)
            c_labels = torch.LongTensor([g.label for g in batch]).to(dev)
            d_labels = torch.cat([torch.ones(batch_size * n, 1), torch.zeros(batch_size * n, 1)], 0).to(dev)
            loss = c_criterion(c_logit, c_labels) + beta * d_criterion(d_logit, d_labels)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            losses.append(float(loss.detach().cpu().numpy()))          # the per-step sync of main.py:43
        scheduler.step()
    assert np.isfinite(losses).all()
    assert np.mean(losses[-8:]) < 0.5 * np.mean(losses[:8])

    # evaluation: one graph per forward, eval mode (main.py:49-57)
    model.eval()
    outs = []
    with torch.no_grad():
        for g in test_graphs:
            c_logit, d_logit = model([g])
            assert c_logit.shape == (1, 2) and d_logit.shape == (2 * n, 1)
            outs.append(c_logit)
    pred = torch.cat(outs, 0).max(1)[1].cpu().numpy()
    acc = float((pred == np.array([g.label for g in test_graphs])).mean())
    assert acc >= 0.8, acc

    # batched evaluation gives the per-graph results (eval-mode BatchNorm: no cross-graph coupling)
    batched = model.predict(test_graphs, batch_size=5)
    assert batched.shape == (len(test_graphs), 2)
    assert float((batched - torch.cat(outs, 0)).abs().max()) <= 1e-5 * float(torch.cat(outs, 0).abs().max())
    assert not model.training
    latent = np.concatenate([model([g], latent=True) for g in test_graphs], 0)      # main.py:71-82
    assert np.abs(model.predict(test_graphs, batch_size=7, latent=True) - latent).max() <= 1e-5 * np.abs(latent).max()
    assert latent.shape == (len(test_graphs), 3 * 32) and np.isfinite(latent).all()
    sal = np.stack([model.compute_saliency([g], 1).detach().cpu().numpy() for g in test_graphs[:3]], 0)
    assert sal.shape == (3, n, 7) and np.isfinite(sal).all() and np.abs(sal).max() > 0

    # checkpoint round trip with the reference's keys (main.py:166)
    path = tmp_path / "model.pt"
    torch.save(model.state_dict(), path)
    model2 = GIN_InfoMaxReg(3, 2, 7, 32, 2, 0.5, True, "sum", "sum", dev).to(dev)
    model2.load_state_dict(torch.load(path, weights_only=True))
    model2.eval()
    with torch.no_grad():
        a, _ = model([test_graphs[0]])
        b, _ = model2([test_graphs[0]])
    assert torch.equal(a, b)
