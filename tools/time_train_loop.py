#!/usr/bin/env python3
"""Wall time of the reference's own training-loop body (main.py:24-43: model(batch); torch losses; zero_grad; backward;
Adam step; the per-step .cpu() sync) at the reference's default batch 32 on 400-node graphs, with and without the
train-mode hipGraph replay of models/graphcnn.py:  python tools/time_train_loop.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
from gnm import synth
from models.graphcnn import GIN_InfoMaxReg
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pool = synth.make_pool("dense_fc", 256, n=400, f0=7)
n = 400
for replay in (False, True):
    torch.manual_seed(0); np.random.seed(0)
    model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).train()
    model.train_replay = replay
    opt = torch.optim.Adam(model.parameters(), lr=0.005)
    ce, bce = torch.nn.CrossEntropyLoss(), torch.nn.BCEWithLogitsLoss()
    d_lab = torch.cat([torch.ones(B * n, 1), torch.zeros(B * n, 1)], 0).to(dev)
    def step(sync=True):
        sel = np.random.permutation(len(pool))[:B]
        batch = [pool[i] for i in sel]
        c, d = model(batch)
        lab = torch.LongTensor([g.label for g in batch]).to(dev)
        loss = ce(c, lab) + 0.05 * bce(d, d_lab)
        opt.zero_grad(); loss.backward(); opt.step()
        return float(loss.detach().cpu().numpy()) if sync else None       # main.py:43
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 100
    for _ in range(K): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(K): step(sync=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("B=%d train_replay=%s: %.3f ms/step with the per-step sync of main.py:43 (%.0f graphs/s), %.3f ms/step without"
          % (B, replay, 1e3 * (t1 - t0) / K, B * K / (t1 - t0), 1e3 * (t2 - t1) / K), flush=True)
