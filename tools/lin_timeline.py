#!/usr/bin/env python3
"""In-kernel timeline of the Linear kernels at the headline shape (tuning build with -DGNM_LIN_TUNING):
    python tools/build_variant.py lintune -DGNM_LIN_TUNING
    GNM_HIP_LIB=graph-neural-mapping_amd/lib/variants/lintune.so python tools/lin_timeline.py [--kernel fwd|bwd|bwd_first]
Lane 0 of every wave stores s_memtime at fixed points.  The counter is per XCD and its rate is not the shader clock:
only differences inside one workgroup are used, and phases are reported as shares of the wave's lifetime next to the
HIP-event duration of the same launch.

fwd (gnm_lin_stream_kernel / gnm_lin_fast_kernel): 0 entry, 1 weight staged, per tile t: 2+4t start, 3+4t operands in
  registers (X staged, A fragments read, next tile's loads issued), 4+4t MFMAs done, 5+4t tile stored; 62 loop left.
bwd (gnm_linear_bwd_fused_kernel): 0 entry, 1 weight staged, per tile t: 2+5t start, 3+5t dZ tile staged (G and Z
  arrived) and X loads issued, 4+5t dgrad MFMAs done, 5+5t wgrad MFMAs done, 6+5t dX stored; 62 loop left, 63 exit
  (dW / BatchNorm partials written).
fwd_split (gnm_lin_split_kernel, the K = H = 64 forward a step runs; build with -DGNM_LIN_TUNING -DGNM_L64_SPLIT_AHEAD so that
  the splits sit in front of the MFMAs): per tile t: 2+6t start, 3+6t staged, 4+6t A planes split, 5+6t MFMAs done,
  6+6t output image written, 7+6t stored; 62 loop left."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np
import torch

from gnm import core

ap = argparse.ArgumentParser()
ap.add_argument("--kernel", default="fwd", choices=["fwd", "fwd_split", "bwd", "bwd_first", "rz", "rz_first"])
args = ap.parse_args()
lib = core.lib
lib.gnm_debug_set_lin_stamps.argtypes = [C.c_void_p]
lib.gnm_debug_set_lin_stamps.restype = None
dev = torch.device("cuda:0")
N, K, H = 409600, 64, 64
f32 = dict(dtype=torch.float32, device=dev)
x, g, z, zlo = torch.randn(N, K, **f32), torch.randn(N, H, **f32), torch.randn(N, H, **f32), torch.randn(N, K, **f32)
W, b = torch.randn(H, K, **f32) * 0.1, torch.randn(H, **f32)
out, dA = torch.empty(N, H, **f32), torch.empty(N, K, **f32)
sc, sh = torch.rand(K, **f32) + 0.5, torch.randn(K, **f32) * 0.1
mean, rstd, cA, m1, m2 = (torch.randn(H, **f32) * 0.1, torch.rand(H, **f32) + 0.5, torch.rand(H, **f32),
                          torch.randn(H, **f32) * 0.01, torch.randn(H, **f32) * 0.01)
lmean, lrstd = torch.randn(K, **f32) * 0.1, torch.rand(K, **f32) + 0.5
dW, db = torch.empty(H, K, **f32), torch.empty(H, **f32)
st = torch.cuda.current_stream().cuda_stream
fwd = args.kernel in ("fwd", "fwd_split")
split = args.kernel == "fwd_split"       # gnm_lin_split_kernel (build with -DGNM_LIN_TUNING -DGNM_L64_SPLIT_AHEAD): 6 stamps per tile
rz = args.kernel.startswith("rz")
grid = lib.gnm_linear_grid(N) if fwd else lib.gnm_linear_bwd_grid(N)
stats = torch.empty(grid, 2, H, dtype=torch.float64, device=dev)
ws = torch.empty(int(lib.gnm_linear_bwd_workspace_floats(N, H, K)), **f32)
lp = torch.empty(grid, 2, K, dtype=torch.float64, device=dev)


def run():
    if fwd:
        core.check(lib.gnm_linear_fwd(x.data_ptr(), K, W.data_ptr(), K, 0, b.data_ptr(), out.data_ptr(), H, N, K, H,
                                      sc.data_ptr(), sh.data_ptr(), 1, stats.data_ptr(), st), "lin")
        return
    second = args.kernel in ("bwd", "rz")
    if rz:
        core.check(lib.gnm_linear_bwd_fused_rz(
            g.data_ptr(), H, b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), cA.data_ptr(), m1.data_ptr(), m2.data_ptr(),
            (zlo if second else x).data_ptr(), K, sc.data_ptr() if second else None, sh.data_ptr() if second else None,
            1 if second else 0, W.data_ptr(), K, dA.data_ptr(), K, dW.data_ptr(), K, db.data_ptr(), ws.data_ptr(), N, K, H,
            zlo.data_ptr() if second else None, K if second else 0, sc.data_ptr() if second else None,
            sh.data_ptr() if second else None, lmean.data_ptr() if second else None, lrstd.data_ptr() if second else None,
            lp.data_ptr() if second else None, st), "rz")
        return
    core.check(lib.gnm_linear_bwd_fused(
        g.data_ptr(), H, z.data_ptr(), H, mean.data_ptr(), rstd.data_ptr(), cA.data_ptr(), m1.data_ptr(), m2.data_ptr(),
        (zlo if second else x).data_ptr(), K, sc.data_ptr() if second else None, sh.data_ptr() if second else None,
        1 if second else 0, W.data_ptr(), K, dA.data_ptr(), K, dW.data_ptr(), K, db.data_ptr(), ws.data_ptr(), N, K, H,
        zlo.data_ptr() if second else None, K if second else 0, sc.data_ptr() if second else None,
        sh.data_ptr() if second else None, lmean.data_ptr() if second else None, lrstd.data_ptr() if second else None,
        lp.data_ptr() if second else None, st), "bwd")


for _ in range(5):
    run()
WPB = 8 if rz else 4                                              # waves per workgroup
nwg = (grid + 1) // 2 if rz else grid
stamps = torch.zeros(nwg * WPB * 64, dtype=torch.int64, device=dev)
lib.gnm_debug_set_lin_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize(); stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.gnm_debug_set_lin_stamps(None)
ms = e0.elapsed_time(e1)
s = stamps.cpu().numpy().reshape(nwg, WPB, 64).astype(np.float64)
per = 6 if split else (4 if fwd else (6 if rz else 5))
names = (["wait for X + prologue + stage (LDS writes, fence)", "next tile's loads + read A image + split (3 x 4 planes)",
          "48 MFMAs (+ B planes from LDS)", "bias + statistics + output image (LDS writes, fence)", "read image + stores"] if split else
         ["wait for X + stage + read A", "64 MFMAs (+ B from LDS)", "epilogue + stores"] if fwd else
         ["wait for X rows + split + Z MFMAs", "wait for G + dZ + image + issue X column loads", "dgrad (image rows, split, MFMAs)",
          "mask / statistics / staging + dX stores", "next tile's requests + wgrad (split, MFMAs)"] if rz else
         ["wait for G, Z + dZ tile to LDS + issue X loads", "dgrad: 64 MFMAs", "wgrad: wait for X + 64 MFMAs",
          "mask / statistics / dX stores"])
NT = 10 if (rz or split) else 12
ntile = (s[:, :, 2:2 + per * NT:per] > 0).sum(2)                  # tiles stamped per wave (first NT at most)
life = np.where(s[:, :, 62] > s[:, :, 0], s[:, :, 62] - s[:, :, 0], np.nan)
print("%s: launch %.1f us (stamped build), grid %d, tiles per wave %d..%d" % (args.kernel, ms * 1e3, grid, ntile.min(), ntile.max()))
print("  shares of a wave's lifetime (entry -> tile loop left), mean over waves:")
print("    weight staging + barrier                        %5.1f %%" % (100 * np.nanmean((s[:, :, 1] - s[:, :, 0]) / life)))
tot = np.zeros((len(names) + 1,) + life.shape)
for t in range(NT):
    ok = s[:, :, 2 + per * t] > 0
    for k in range(len(names)):
        tot[k] += np.where(ok, s[:, :, 3 + per * t + k] - s[:, :, 2 + per * t + k], 0.0)
    if t + 1 < NT:
        ok2 = ok & (s[:, :, 2 + per * (t + 1)] > 0)
        tot[-1] += np.where(ok2, s[:, :, 2 + per * (t + 1)] - s[:, :, 2 + per * t + len(names)], 0.0)
for k, nm in enumerate(names):
    print("    %-47s %5.1f %%" % (nm, 100 * (tot[k] / life).mean()))
print("    %-47s %5.1f %%" % ("between tiles", 100 * (tot[-1] / life).mean()))
last = s[:, :, 62].max(1, keepdims=True)
print("    idle until the block's last wave leaves the loop  %5.1f %% (in addition)" % (100 * np.nanmean((last - s[:, :, 62]) / life)))
if not fwd:
    print("    loop left -> kernel exit (dW / statistics combine)  %5.1f %% (in addition)" % (100 * np.nanmean((s[:, :, 63] - s[:, :, 62]) / life)))
blk = s[:, :, 62].max(1) - s[:, :, 0].min(1)
print("  block lifetime spread (ticks): min %.0f median %.0f max %.0f" % (blk.min(), np.median(blk), blk.max()))
