#!/usr/bin/env python3
"""Golden TRAINING TRAJECTORY from the real reference model (build container only).

    python tests/golden/make_train_golden.py

Drives /root/reference's GIN_InfoMaxReg (graphcnn.py:12) through the steps of the reference's
train() (main.py:19-47) -- forward, CrossEntropyLoss + beta * BCEWithLogitsLoss with d_labels sized by node
count, zero_grad, backward, Adam step -- and its StepLR (main.py:137,153), on seeded synthetic graphs, and
stores DATA ONLY: the per-step losses, the parameters/buffers after the last step, and Adam's moments.
Dropout is 0 so the trajectory is deterministic; one np.random.permutation(B) per forward (graphcnn.py:199)
comes from np.random.seed(NP_SEED + step).
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_goldens as mg  # reuses the graph generator and the seeded model builder (imports the reference)

STEPS, LR, BETA, STEP_SIZE, GAMMA = 6, 0.01, 0.05, 2, 0.5


def main():
    L, m, f0, H, C, B, n, t = 3, 2, 5, 32, 2, 4, 20, 40
    graphs, raw = mg.make_batch(700, B, n, t, f0)
    model = mg.build_model(1, L, m, f0, H, C, 0.0, True, "sum", "sum")
    sd0 = np.load(os.path.join(HERE, "state_seed1.npz"))
    for k, v in model.state_dict().items():
        assert np.array_equal(sd0[k], v.numpy()), k          # same start as the tiny_s1 cases
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=STEP_SIZE, gamma=GAMMA)
    ce, bce = torch.nn.CrossEntropyLoss(), torch.nn.BCEWithLogitsLoss()
    labels = torch.LongTensor([g.label for g in graphs])
    N = B * n
    d_labels = torch.cat([torch.ones(N, 1), torch.zeros(N, 1)], 0)
    out = {"cfg": np.array([L, m, f0, H, C, 1, B, n], dtype=np.int64), "hyper": np.array([LR, BETA, STEP_SIZE, GAMMA]),
           "steps": np.array(STEPS), "np_seed": np.array(mg.NP_SEED), "labels": labels.numpy()}
    for g, (und, feats, label) in enumerate(raw):
        out[f"und_{g}"] = und.astype(np.int16)
        out[f"feat_{g}"] = feats
    losses, perms, lrs = [], [], []
    model.train()
    for s in range(STEPS):
        np.random.seed(mg.NP_SEED + s)
        state = np.random.get_state()
        perms.append(np.random.permutation(B))
        np.random.set_state(state)
        c_logit, d_logit = model(graphs)
        c_loss, d_loss = ce(c_logit, labels), bce(d_logit, d_labels)
        loss = c_loss + BETA * d_loss
        lrs.append(opt.param_groups[0]["lr"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()                       # per "epoch" in main.py; per step here to exercise lr changes
        losses.append([loss.item(), c_loss.item(), d_loss.item()])
    out["losses"] = np.array(losses, dtype=np.float64)
    out["perms"] = np.array(perms, dtype=np.int64)
    out["lrs"] = np.array(lrs, dtype=np.float64)
    for k, v in model.state_dict().items():
        out["final_" + k] = v.detach().numpy().copy()
    for name, p in model.named_parameters():
        st = opt.state[p]
        out["expavg_" + name] = st["exp_avg"].numpy().copy()
        out["expavgsq_" + name] = st["exp_avg_sq"].numpy().copy()
    path = os.path.join(HERE, "train_s1.npz")
    np.savez_compressed(path, **out)
    print("train_s1: %d KB" % (os.path.getsize(path) / 1024), np.array(losses)[:, 0])


if __name__ == "__main__":
    main()
