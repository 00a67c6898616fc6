#!/usr/bin/env python3
"""Host profile of one-graph-per-forward evaluation (main.py:49-57): where the Python time of model([g]) goes."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
from gnm import synth
from models.graphcnn import GIN_InfoMaxReg
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).eval()
graphs = synth.make_pool("dense_fc", 64)
model.arena().add_many(graphs)          # steady state: main.py evaluates the same graphs every epoch
with torch.no_grad():
    for g in graphs[:8]:
        model([g])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for k in range(480):
        model([graphs[k % 64]])
    pr.disable()
    print("host %.3f ms per forward (480 forwards, no sync inside)" % ((time.perf_counter() - t0) / 480 * 1e3))
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
