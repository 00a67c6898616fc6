#!/usr/bin/env python3
"""Micro-benchmark of the Linear kernels at the headline shape (N = 409,600 rows, K = H = 64) through the C-ABI:
  fwd        gnm_linear_fwd with the BatchNorm+ReLU prologue and the column statistics (what a step runs)
  bwd        gnm_linear_bwd_fused, second Linear of an MLP (also reduces the BatchNorm below it)
  bwd_first  gnm_linear_bwd_fused, first Linear of an MLP (its input is the aggregation output: no prologue)
--ab a.so,b.so: other builds of the library timed INTERLEAVED with the loaded one (paired differences)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np
import torch

from gnm import _cabi, core


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=409600)
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--H", type=int, default=64)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--modes", default="fwd,bwd,bwd_first")
    ap.add_argument("--ab", default=None)
    ap.add_argument("--check", action="store_true",
                    help="fwd only: every build's output and statistics, over 10 launches each, must equal the loaded "
                         "build's first launch bit for bit")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N, K, H = args.N, args.K, args.H
    libs = [(os.path.basename(os.environ.get("GNM_HIP_LIB", "product")), core.lib)]
    for path in (args.ab.split(",") if args.ab else []):
        L = C.CDLL(path)
        for name, (res, at) in _cabi.SIGNATURES.items():
            if hasattr(L, name):
                getattr(L, name).restype = res
                getattr(L, name).argtypes = at
        libs.append((os.path.basename(path), L))
    f32 = dict(dtype=torch.float32, device=dev)
    x, g, z, zlo = (torch.randn(N, K, **f32), torch.randn(N, H, **f32), torch.randn(N, H, **f32), torch.randn(N, K, **f32))
    W, b = torch.randn(H, K, **f32) * 0.1, torch.randn(H, **f32)
    out, dA = torch.empty(N, H, **f32), torch.empty(N, K, **f32)
    sc, sh = torch.rand(K, **f32) + 0.5, torch.randn(K, **f32) * 0.1
    mean, rstd, cA, m1, m2 = (torch.randn(H, **f32) * 0.1, torch.rand(H, **f32) + 0.5, torch.rand(H, **f32), torch.randn(H, **f32) * 0.01,
                              torch.randn(H, **f32) * 0.01)
    lmean, lrstd = torch.randn(K, **f32) * 0.1, torch.rand(K, **f32) + 0.5
    dW, db = torch.empty(H, K, **f32), torch.empty(H, **f32)
    st = torch.cuda.current_stream().cuda_stream

    def run(mode, L):
        if mode == "fwd":
            grid = L.gnm_linear_grid(N)
            stats = run.bufs.setdefault(("stats", grid), torch.empty(grid, 2, H, dtype=torch.float64, device=dev))
            core.check(L.gnm_linear_fwd(x.data_ptr(), K, W.data_ptr(), K, 0, b.data_ptr(), out.data_ptr(), H, N, K, H,
                                        sc.data_ptr(), sh.data_ptr(), 1, stats.data_ptr(), st), "fwd")
            return
        ws = run.bufs.setdefault(("ws", id(L)), torch.empty(int(L.gnm_linear_bwd_workspace_floats(N, H, K)), **f32))
        second = mode == "bwd"
        lp = run.bufs.setdefault(("lp", id(L)), torch.empty(L.gnm_linear_bwd_grid(N), 2, K, dtype=torch.float64, device=dev))
        core.check(L.gnm_linear_bwd_fused(
            g.data_ptr(), H, z.data_ptr(), H, mean.data_ptr(), rstd.data_ptr(), cA.data_ptr(), m1.data_ptr(), m2.data_ptr(),
            (zlo if second else x).data_ptr(), K, sc.data_ptr() if second else None, sh.data_ptr() if second else None,
            1 if second else 0, W.data_ptr(), K, dA.data_ptr(), K, dW.data_ptr(), K, db.data_ptr(), ws.data_ptr(), N, K, H,
            zlo.data_ptr() if second else None, K if second else 0, sc.data_ptr() if second else None,
            sh.data_ptr() if second else None, lmean.data_ptr() if second else None, lrstd.data_ptr() if second else None,
            lp.data_ptr() if second else None, st), "bwd")
    run.bufs = {}

    for mode in args.modes.split(","):
        for _, L in libs:
            for _ in range(4):
                run(mode, L)
        torch.cuda.synchronize()
        evs = {name: [] for name, _ in libs}
        for _ in range(args.iters):
            for name, L in libs:
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(); run(mode, L); eb.record()
                evs[name].append((ea, eb))
        torch.cuda.synchronize()
        byt = 4.0 * N * (K + H) * (1 if mode == "fwd" else 2)
        flop = 2.0 * N * K * H * (1 if mode == "fwd" else 2)
        base = None
        for name, _ in libs:
            ms = np.array([a.elapsed_time(b_) for a, b_ in evs[name]])
            med = float(np.median(ms))
            rel = "" if base is None else "   paired vs %s: %+.1f us" % (libs[0][0], 1e3 * float(np.median(ms - base)))
            if base is None:
                base = ms
            print("LIN %-14s %-9s N=%d K=%d H=%d: median %.1f us (min %.1f)  %.0f GB/s  %.1f TFLOP/s%s"
                  % (name, mode, N, K, H, med * 1e3, ms.min() * 1e3, byt / med / 1e6, flop / med / 1e9, rel), flush=True)


    if args.check:
        want = torch.relu(x.double() * sc.double() + sh.double()) @ W.double().t() + b.double()
        ref = None
        for name, L in libs:
            bad = wrong = 0
            for it in range(10):
                out.fill_(float("nan"))
                run("fwd", L)
                stats = run.bufs[("stats", L.gnm_linear_grid(N))]
                cur = (out.clone(), stats.sum(0).clone())
                err = (cur[0].double() - want).abs().amax(1)          # per row
                rows = (~(err < 1e-3)).nonzero().flatten()
                if rows.numel():
                    wrong += 1
                    if wrong == 1:
                        tiles = torch.unique(rows // 32)
                        r = int(rows[0])
                        print("   launch %d: %d wrong rows in %d tiles; tiles %s; rows-in-tile of the first: %s" % (
                            it, rows.numel(), tiles.numel(), tiles[:16].tolist(), (rows[rows // 32 == r // 32] % 32).tolist()))
                        print("   row %d got %s\n           want %s" % (r, cur[0][r, :6].tolist(), want[r, :6].tolist()))
                        cols = ((cur[0][r].double() - want[r]).abs() > 1e-3).nonzero().flatten()
                        print("   wrong columns of row %d: %s" % (r, cols.tolist()))
                        t0 = r // 32
                        blk = (cur[0][32 * t0:32 * t0 + 32].double() - want[32 * t0:32 * t0 + 32]).abs() > 1e-3
                        for rr in range(32):
                            if blk[rr].any():
                                cc = blk[rr].nonzero().flatten().tolist()
                                iv = cur[0][32 * t0 + rr].view(torch.int32)[cc].tolist()
                                print("     tile %d row %2d: cols %s  as int32 %s" % (t0, rr, cc, iv))
                        c0 = int(cols[0])
                        seg = cur[0][r, c0:c0 + 4]
                        print("   got %s want %s" % (seg.tolist(), want[r, c0:c0 + 4].tolist()))
                        dd = (want[:, c0:c0 + 4].float() - seg).abs().amax(1)
                        print("   that 16-B piece is the true result of row %d (err %.2e)" % (int(dd.argmin()), float(dd.min())))
                        xin = torch.relu(x * sc + sh)
                        for k0 in range(0, K - 3):
                            dx = (xin[:, k0:k0 + 4] - seg).abs().amax(1)
                            if float(dx.min()) < 1e-6:
                                print("   ... or the staged INPUT of row %d, columns %d.." % (int(dx.argmin()), k0))
                if ref is None:
                    ref = cur
                bad += int(not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])))
            print("CHECK %-14s fwd: %d of 10 launches have rows off the fp64 product by > 1e-3; %d differ bitwise from %s's first"
                  % (name, wrong, bad, libs[0][0]), flush=True)


if __name__ == "__main__":
    main()
