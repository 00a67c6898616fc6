#!/usr/bin/env python3
"""Stand-alone timing of the discriminator score kernel (csrc/disc.hip, the UNIT form a training step launches) through
the C-ABI: HIP events, medians, over a range of graph counts -- is the launch linear in B, or does it step where the
workgroups stop fitting the chip in one round?   python tools/bench_disc.py [--n 400] [--H 64] [--L 5] [--plain]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np
import torch
from gnm._cabi import check, lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400)
ap.add_argument("--H", type=int, default=64)
ap.add_argument("--L", type=int, default=5)
ap.add_argument("--plain", action="store_true", help="gnm_disc_score_fwd (no backward by-products)")
ap.add_argument("--batches", default="256,512,768,1024,1536,2048")
args = ap.parse_args()
dev = torch.device("cuda:0")
n, H, L = args.n, args.H, args.L
st = torch.cuda.current_stream().cuda_stream
for B in [int(b) for b in args.batches.split(",")]:
    N, LH = B * n, L * H
    zs = [torch.randn(N, H, device=dev) for _ in range(L)]
    sc = [torch.rand(H, device=dev) + 0.5 for _ in range(L)]
    sh = [torch.randn(H, device=dev) * 0.3 for _ in range(L)]
    U = torch.randn(B, LH, device=dev) * 0.05
    node_off = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
    perm = torch.randperm(B, device=dev).to(torch.int32)
    bias = torch.tensor([0.3], device=dev)
    hp = (C.c_void_p * L)(*[z.data_ptr() for z in zs])
    sp = (C.c_void_p * L)(*[s.data_ptr() for s in sc])
    tp = (C.c_void_p * L)(*[s.data_ptr() for s in sh])
    d = torch.empty(2 * N, device=dev)
    ldunit = (LH + 2 + 3) & ~3
    unit = torch.empty(B, ldunit, device=dev)
    ip = torch.empty(B, dtype=torch.int32, device=dev)

    def run():
        if args.plain:
            check(lib.gnm_disc_score_fwd(hp, sp, tp, H, L, H, U.data_ptr(), LH, perm.data_ptr(), bias.data_ptr(),
                                         node_off.data_ptr(), N, B, d.data_ptr(), st), "fwd")
        else:
            check(lib.gnm_disc_score_fwd_unit(hp, sp, tp, H, L, H, U.data_ptr(), LH, perm.data_ptr(), bias.data_ptr(),
                                              node_off.data_ptr(), N, B, d.data_ptr(), unit.data_ptr(), ldunit,
                                              ip.data_ptr(), st), "fwd unit")
    for _ in range(5):
        run()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = float(np.median(ts))
    byt = 4.0 * N * LH + 8.0 * N
    print("B=%5d  %7.1f us  %6.0f GB/s  %.3f us per graph" % (B, us, byt / us / 1e3, us / B))
