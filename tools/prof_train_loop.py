#!/usr/bin/env python3
"""Host profile (cProfile) of the reference-shaped training loop body at batch 32 with the train-mode replay on:
how much of a step is this package's glue, how much the caller's own torch losses / optimizer."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
from gnm import synth
from models.graphcnn import GIN_InfoMaxReg
dev = torch.device("cuda:0")
B, n = 32, 400
pool = synth.make_pool("dense_fc", 128, n=400, f0=7)
torch.manual_seed(0); np.random.seed(0)
model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=0.005)
ce, bce = torch.nn.CrossEntropyLoss(), torch.nn.BCEWithLogitsLoss()
d_lab = torch.cat([torch.ones(B * n, 1), torch.zeros(B * n, 1)], 0).to(dev)
T = {"fwd": 0.0, "loss": 0.0, "zero": 0.0, "bwd": 0.0, "step": 0.0}
def step(timed=False):
    sel = np.random.permutation(len(pool))[:B]
    batch = [pool[i] for i in sel]
    t0 = time.perf_counter()
    c, d = model(batch)
    t1 = time.perf_counter()
    lab = torch.LongTensor([g.label for g in batch]).to(dev)
    loss = ce(c, lab) + 0.05 * bce(d, d_lab)
    t2 = time.perf_counter()
    opt.zero_grad()
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    if timed:
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            T[k] += v
for _ in range(10): step()
torch.cuda.synchronize()
K = 100
t0 = time.perf_counter()
for _ in range(K): step(True)
torch.cuda.synchronize()
print("%.3f ms/step (no per-step sync); host time per phase (ms): %s" % (1e3 * (time.perf_counter() - t0) / K,
      {k: round(1e3 * v / K, 3) for k, v in T.items()}), flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
