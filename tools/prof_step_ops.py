#!/usr/bin/env python3
"""Which torch ops (not libgnm_hip.so launches) does one eager training step of the bench configuration still issue?
torch.profiler over 3 eager steps at B = 1024: CUDA-kernel-launching aten ops with their counts per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from gnm import synth
from gnm.parallel import DataParallelGIN
from gnm.train import infomax_loss
from models.graphcnn import GIN_InfoMaxReg
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pool = synth.make_pool("dense_fc", B, n=400, f0=7)
torch.manual_seed(0)
model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).train()
dp = DataParallelGIN(model)
arena = model.arena()
gids = np.array(arena.add_many(pool), dtype=np.int64)
labels = torch.tensor([g.label for g in pool], device=dev)
bt = arena.batch_from_gids(gids)
def step():
    dp.zero_grad()
    c, d = model.forward_batch(bt, X=arena.features(bt), perm=np.random.permutation(B))
    infomax_loss(c, d, labels, 0.05)[0].backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.device_time_total > 0 or "aten::" in e.key]
rows.sort(key=lambda e: -e.count)
print("%-60s %6s %10s" % ("op", "n/step", "dev us/step"))
for e in rows[:60]:
    print("%-60s %6.1f %10.1f" % (e.key[:60], e.count / 3.0, e.device_time_total / 3.0))
