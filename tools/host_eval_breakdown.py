#!/usr/bin/env python3
"""Where the host time of one replayed eval forward goes (model([g]) in eval mode, main.py:49-57): per-call wall time of
the steps of GIN_InfoMaxReg._forward_eval_replay, no GPU sync inside."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
from gnm import synth, graphs as G
from models.graphcnn import GIN_InfoMaxReg
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).eval()
model.eval_fused = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "0" else False
pool = synth.make_pool("dense_fc", 64)
arena = model.arena()
arena.add_many(pool)
for g in pool[:8]:
    model([g])
torch.cuda.synchronize()
T = {}
def timed(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[label] = T.get(label, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, w)
ce = next(iter(model._eval_cache.values()))
timed(arena, "add_many", "arena.add_many")
timed(arena, "refresh_agg0", "arena.refresh_agg0")
timed(ce, "valid_for", "capture.valid_for")
timed(ce.static, "load_gids", "static.load_gids")
timed(ce.graph, "replay", "graph.replay")
K = 400
t0 = time.perf_counter()
for k in range(K):
    model([pool[k % 64]])
tot = time.perf_counter() - t0
torch.cuda.synchronize()
print("eval_fused=%s: %.1f us per forward on the host" % (model.eval_fused, 1e6 * tot / K))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("   %-24s %6.1f us" % (k, 1e6 * v / K))
print("   %-24s %6.1f us" % ("everything else", 1e6 * (tot - sum(T.values())) / K))
