#!/usr/bin/env python3
"""What the GPU box gives the job in host cores, and what the torch-CPU baseline step costs at different thread counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try:
        print(p, open(p).read().strip(), flush=True)
    except Exception as e:
        print(p, "n/a", flush=True)
from gnm import synth
from oracle import gin_oracle as O
from oracle.gin_torch_cpu import TorchCpuGIN
pool = synth.make_pool("dense_fc", 32, first=0, n=400, f0=7, keep_pct=30.0)
state = dict(np.load(os.path.join(ROOT, "tests", "golden", "state_seed0.npz")))
ob = [O.OGraph(len(g.g), g.edge_mat.numpy(), g.node_features.numpy(), g.label) for g in pool]
perm = np.arange(32)
for th in (16, 32, 64):
    if th > (os.cpu_count() or 1):
        continue
    torch.set_num_threads(th)
    m = TorchCpuGIN(state, 5, 2, True, "sum", "sum")
    for want in (True, False):
        t0 = time.perf_counter(); m.train_step(ob, perm, 0.05, want); t1 = time.perf_counter()
        m.train_step(ob, perm, 0.05, want); t2 = time.perf_counter()
        print("threads %d want_disc %s: first %.2f s, second %.2f s" % (th, want, t1 - t0, t2 - t1), flush=True)
