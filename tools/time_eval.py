#!/usr/bin/env python3
"""GPU time per graph of the reference's evaluation pattern -- model([g]) per graph in eval mode (main.py:49-57; the
reference does NOT wrap it in no_grad, the outputs are detached) -- with the one-launch encoder on / off and with the
call under torch.no_grad(); eval_fused = "layers": one launch per layer, a workgroup per 32-row block:  python tools/time_eval.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
from gnm import synth
from models.graphcnn import GIN_InfoMaxReg
dev = torch.device("cuda:0")
graphs = synth.make_pool("dense_fc", 64)
for fused in (False, True, "layers"):
    for nograd in (False, True):
        torch.manual_seed(0)
        model = GIN_InfoMaxReg(5, 2, 7, 64, 2, 0.5, True, "sum", "sum", dev).to(dev).eval()
        model.eval_fused = fused
        model.arena().add_many(graphs)
        ctx = torch.no_grad() if nograd else torch.enable_grad()
        with ctx:
            for g in graphs[:8]:
                model([g])[0].detach()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 200
            for k in range(K):
                model([graphs[k % 64]])[0].detach()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print("eval_fused=%s no_grad=%s: %.3f ms per graph end to end (host enqueue %.3f ms)"
              % (fused, nograd, 1e3 * (t2 - t0) / K, 1e3 * (t1 - t0) / K), flush=True)
