#!/usr/bin/env python3
"""The tail's three [B, L*H]-sized products: csrc/sgemm.hip against torch.mm (hipBLASLt), HIP-event medians of single
launches at the headline shape (B = 1024, L*H = 320) and at B = 32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
from gnm import core
lib = core.lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(1 << 26, device=dev)


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    evs = []
    for _ in range(iters):
        flush.add_(1.0)                       # other work between launches, as in the step
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return 1e3 * float(np.median([a.elapsed_time(b) for a, b in evs]))


for B in (1024, 32):
    LH = 320
    c, dU, Wd = torch.randn(B, LH, device=dev), torch.randn(B, LH, device=dev), torch.randn(LH, LH, device=dev)
    U, T, dW = torch.empty(B, LH, device=dev), torch.empty(B, LH, device=dev), torch.empty(LH, LH, device=dev)
    cases = [("U = c Wd^T", lambda: lib.gnm_small_gemm(c.data_ptr(), LH, 0, Wd.data_ptr(), LH, 0, U.data_ptr(), LH, B, LH, LH, st),
              lambda: torch.mm(c, Wd.t(), out=U)),
             ("T = dU Wd", lambda: lib.gnm_small_gemm(dU.data_ptr(), LH, 0, Wd.data_ptr(), LH, 1, T.data_ptr(), LH, B, LH, LH, st),
              lambda: torch.mm(dU, Wd, out=T)),
             ("dWd = dU^T c", lambda: lib.gnm_small_gemm(dU.data_ptr(), LH, 1, c.data_ptr(), LH, 1, dW.data_ptr(), LH, LH, LH, B, st),
              lambda: torch.mm(dU.t(), c, out=dW))]
    for name, mine, lib_ in cases:
        print("B = %4d  %-14s sgemm %6.1f us   torch.mm %6.1f us" % (B, name, timeit(mine), timeit(lib_)), flush=True)
