#!/usr/bin/env python3
"""In-kernel timeline of the pipelined Linear forward (tuning build with -DGNM_LIN_TUNING):
    GNM_HIP_LIB=graph-neural-mapping_amd/lib/variants/lintune.so python tools/lin_timeline.py
stamps per wave: 0 entry, 1 weight staged (barrier), per tile t: 2+4t start, 3+4t operands in registers (X staged,
A fragments read, next tile's loads issued), 4+4t MFMAs done, 5+4t tile stored; 62 tile loop left (the BatchNorm-statistics
combine that follows is not stamped)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np
import torch

from gnm import core

lib = core.lib
lib.gnm_debug_set_lin_stamps.argtypes = [C.c_void_p]
lib.gnm_debug_set_lin_stamps.restype = None
dev = torch.device("cuda:0")
N, K, H = 409600, 64, 64
x = torch.randn(N, K, device=dev)
W = torch.randn(H, K, device=dev) * 0.1
b = torch.randn(H, device=dev)
z = torch.empty(N, H, device=dev)
sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
grid = lib.gnm_linear_grid(N)
stats = torch.empty(grid, 2, H, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    core.check(lib.gnm_linear_fwd(x.data_ptr(), K, W.data_ptr(), K, 0, b.data_ptr(), z.data_ptr(), H, N, K, H,
                                  sc.data_ptr(), sh.data_ptr(), 1, stats.data_ptr(), st), "lin")


for _ in range(5):
    run()
stamps = torch.zeros(grid * 4 * 64, dtype=torch.int64, device=dev)
lib.gnm_debug_set_lin_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize(); stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.gnm_debug_set_lin_stamps(None)
ms = e0.elapsed_time(e1)
s = stamps.cpu().numpy().reshape(grid, 4, 64).astype(np.float64)
wave_ticks = (s[:, :, 62] - s[:, :, 0])
blk_ticks = s[:, :, 62].max(1) - s[:, :, 0].min(1)
nt = int((s[0, 0, 2:62:4] > 0).sum())
blocks_per_cu = grid / 256.0
print("launch %.1f us (stamped), grid %d (%.1f blocks per CU resident together), %d tiles per wave" % (ms * 1e3, grid, blocks_per_cu, nt))
ghz = blk_ticks.mean() / (ms * 1e6)       # all blocks co-resident: a block lives for (almost) the whole launch
us = lambda t: t / (ghz * 1e3)
print("clock ~ %.2f GHz (block lifetime = launch)" % ghz)
print("  weight staging + barrier     %6.2f us" % us((s[:, :, 1] - s[:, :, 0]).mean()))
tot = {"stage": 0.0, "mfma": 0.0, "store": 0.0, "gap": 0.0}
for t in range(nt):
    tot["stage"] += (s[:, :, 3 + 4 * t] - s[:, :, 2 + 4 * t]).mean()
    tot["mfma"] += (s[:, :, 4 + 4 * t] - s[:, :, 3 + 4 * t]).mean()
    tot["store"] += (s[:, :, 5 + 4 * t] - s[:, :, 4 + 4 * t]).mean()
    if t + 1 < nt:
        tot["gap"] += (s[:, :, 2 + 4 * (t + 1)] - s[:, :, 5 + 4 * t]).mean()
print("  per wave, over its %d tiles:" % nt)
print("    wait for X + stage + read A  %6.2f us  (%.2f per tile)" % (us(tot["stage"]), us(tot["stage"]) / nt))
print("    64 MFMAs (+ B from LDS)      %6.2f us  (%.2f per tile; 64 x 64 cycles alone = %.2f)" % (us(tot["mfma"]), us(tot["mfma"]) / nt, 4096 / (ghz * 1e3)))
print("    epilogue + stores            %6.2f us  (%.2f per tile)" % (us(tot["store"]), us(tot["store"]) / nt))
print("  wave total                   %6.2f us ; idle wrt block's last wave %.2f us" % (us(wave_ticks.mean()), us((s[:, :, 62].max(1, keepdims=True) - s[:, :, 62]).mean())))
