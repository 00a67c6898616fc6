#!/usr/bin/env python3
"""In-kernel timeline of the 64-wide aggregation kernel (tuning build only):
    GNM_HIP_LIB=graph-neural-mapping_amd/lib/variants/tuning.so python tools/agg_timeline.py [--mode plain|fused|bwdstats]
Every wave stamps s_memtime at: 0 kernel entry, 1 own phase A done, 2 its memory queue drained, 3 past the barrier,
then per group k: 4+5k start, 5+5k gather done, 6+5k combined, 7+5k epilogue math done, 8+5k store issued; 63 gather loop left.  Prints where an
average wave's time goes (us at the clock implied by the launch's HIP-event duration)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from gnm import core, synth
from gnm.arena import GraphArena


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="plain")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--config", default="c2", choices=["c2", "c4"], help="c4: 1000-node kNN graphs, F = 128 (4 slices of 32)")
    args = ap.parse_args()
    lib = core.lib
    lib.gnm_debug_set_stamps.argtypes = [C.c_void_p]
    lib.gnm_debug_set_stamps.restype = None
    dev = torch.device("cuda:0")
    c4 = args.config == "c4"
    if c4:
        args.batch = min(args.batch, 256)
    pool = synth.make_pool("knn", args.batch) if c4 else synth.make_pool("dense_fc", args.batch)
    ar = GraphArena(dev)
    gids = np.array(ar.add_many(pool), dtype=np.int64)
    batch = ar.batch_from_gids(gids)
    N, F, B = batch.N, (128 if c4 else 64), batch.B
    WG = B * (4 if c4 else 1)                 # workgroups per launch (one per graph and slice)
    x, h, y = torch.randn(N, F, device=dev), torch.randn(N, F, device=dev), torch.empty(N, F, device=dev)
    eps = torch.zeros(1, device=dev)
    spec = core.GinSpec(5, 2, True, "sum", "sum")
    part = torch.empty(core.lib.gnm_agg_num_partials(F, batch.n_max, B), dtype=torch.float64, device=dev)
    sc, sh = torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev) * 0.3
    mu, rs = torch.randn(F, device=dev) * 0.1, torch.rand(F, device=dev) + 0.5
    gf, dpool, U = torch.empty(B, F, device=dev), torch.randn(B, F, device=dev), torch.randn(B, F, device=dev)
    dsc1, s2sum = torch.randn(N, device=dev), torch.randn(B, device=dev)
    inv_perm = torch.randperm(B, device=dev).to(torch.int32)
    spart = torch.empty(B, 2, F, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    a = ar

    def run():
        if args.mode == "fused":
            core.check(lib.gnm_agg_fwd_bnrelu(
                a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.rp_off.data_ptr(), batch.col_off.data_ptr(),
                batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max, x.data_ptr(), F, sc.data_ptr(),
                sh.data_ptr(), h.data_ptr(), F, gf.data_ptr(), F, 0, y.data_ptr(), F, F, eps.data_ptr(), 0, 0, st), "f")
        elif args.mode == "bwdstats":
            core.check(lib.gnm_agg_bwd_stats(
                a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(), batch.t_col_off.data_ptr(),
                a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(), batch.node_off.data_ptr(), B, batch.n_max,
                batch.nnz_max, x.data_ptr(), F, y.data_ptr(), F, F, eps.data_ptr(), 0, 0, None, 0, part.data_ptr(),
                h.data_ptr(), F, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), dpool.data_ptr(), F, 0,
                dsc1.data_ptr(), U.data_ptr(), F, inv_perm.data_ptr(), s2sum.data_ptr(), spart.data_ptr(), st), "b")
        else:
            core._agg(batch, x, y, F, eps.data_ptr(), spec, False)

    for _ in range(5):
        run()
    stamps = torch.zeros(WG * 16 * 64, dtype=torch.int64, device=dev)
    lib.gnm_debug_set_stamps(stamps.data_ptr())
    run()
    torch.cuda.synchronize()
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    lib.gnm_debug_set_stamps(None)
    ms = e0.elapsed_time(e1)
    raw = stamps.cpu().numpy().reshape(WG, 16, 64)
    raw = raw[:, (raw[0, :, 0] != 0), :]          # waves that exist (8 per workgroup in the sliced kernel)
    s = raw.astype(np.float64)
    # the counter is per XCD (not synchronised across the chip): only differences inside one workgroup mean anything
    wg_ticks = (s[:, :, 63].max(1) - s[:, :, 0].min(1))
    # clock: 1024 workgroups run back to back on 256 CUs, so (ticks per workgroup) x B / 256 ~ launch duration
    ghz = wg_ticks.mean() * WG / 256.0 / (ms * 1e6)          # ticks per ns (an upper bound: ignores dispatch gaps)
    us = lambda ticks: ticks / (ghz * 1e3)
    print("mode %s: launch %.1f us by HIP events (stamped build); workgroup %.0f ticks -> clock <= %.2f GHz" %
          (args.mode, ms * 1e3, wg_ticks.mean(), ghz))
    ng = int(((s[0, 0, 4:60:5] > 0).sum()))
    if ng == 0:       # the sliced kernel stamps entry / tile loaded / past the barrier / loop left only
        print("per workgroup (mean over %d workgroups x %d waves):" % (WG, raw.shape[1]))
        print("  own phase A (tile load + LDS writes)  %6.2f us" % us((s[:, :, 1] - s[:, :, 0]).mean()))
        print("  barrier wait                           %6.2f us" % us((s[:, :, 3] - s[:, :, 1]).mean()))
        print("  gather + epilogue + stores             %6.2f us" % us((s[:, :, 63] - s[:, :, 3]).mean()))
        print("  idle after own last group              %6.2f us" % us((s[:, :, 63].max(1, keepdims=True) - s[:, :, 63]).mean()))
        print("  workgroup total                        %6.2f us   x %d workgroups / 256 CUs = %.1f us"
              % (us(wg_ticks.mean()), WG, us(wg_ticks.mean()) * WG / 256))
        return
    phaseA = us((s[:, :, 1] - s[:, :, 0]).mean())
    drain = us((s[:, :, 2] - s[:, :, 1]).mean())
    barrier = us((s[:, :, 3] - s[:, :, 2]).mean())
    gather = us(sum((s[:, :, 5 + 5 * k] - s[:, :, 4 + 5 * k]) for k in range(ng)).mean())
    epi = us(sum((s[:, :, 8 + 5 * k] - s[:, :, 5 + 5 * k]) for k in range(ng)).mean())
    epi_parts = [us(sum((s[:, :, j + 1 + 5 * k] - s[:, :, j + 5 * k]) for k in range(ng)).mean()) for j in (5, 6, 7)]
    inter = us((sum((s[:, :, 4 + 5 * (k + 1)] - s[:, :, 8 + 5 * k]) for k in range(ng - 1)) + (s[:, :, 63] - s[:, :, 8 + 5 * (ng - 1)])).mean())
    wave_total = us((s[:, :, 63] - s[:, :, 0]).mean())
    wg_total = us((s[:, :, 63].max(1) - s[:, :, 0].min(1)).mean())
    tail = us((s[:, :, 63].max(1, keepdims=True) - s[:, :, 63]).mean())
    first_group = us((s[:, :, 5] - s[:, :, 4]).mean())
    last_group = us((s[:, :, 5 + 5 * (ng - 1)] - s[:, :, 4 + 5 * (ng - 1)]).mean())
    print("per workgroup (mean over %d workgroups x %d waves), %d groups per wave:" % (WG, raw.shape[1], ng))
    print("  own phase A (tile load)         %6.2f us" % phaseA)
    print("  drain vmcnt(0)                  %6.2f us" % drain)
    print("  barrier wait                    %6.2f us" % barrier)
    print("  gather, all groups              %6.2f us   (first group %.2f, last group %.2f)" % (gather, first_group, last_group))
    print("  combine + epilogue + store      %6.2f us   (combine %.2f, epilogue math %.2f, store issue %.2f)" % ((epi,) + tuple(epi_parts)))
    print("  between groups / loop exit      %6.2f us" % inter)
    print("  wave total                      %6.2f us" % wave_total)
    print("  idle after own last group       %6.2f us   (waiting for the workgroup's slowest wave)" % tail)
    print("  workgroup total                 %6.2f us   x %d workgroups / 256 CUs = %.1f us" % (wg_total, WG, wg_total * WG / 256))
    # start skew of successive workgroups on one CU cannot be seen from here; launch - sum = dispatch gaps
    rows = []
    for k in range(ng):
        rows.append("%.2f" % us((s[:, :, 5 + 5 * k] - s[:, :, 4 + 5 * k]).mean()))
    print("  gather per group                " + " ".join(rows))
    print("  (times are shares of the workgroup at the clock above; ticks per workgroup %.0f)" % wg_ticks.mean())


if __name__ == "__main__":
    main()
