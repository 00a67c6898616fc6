#!/usr/bin/env python3
"""Micro-benchmark of the aggregation kernel alone (forward, F = 64) through the C-ABI:
mean launch time from HIP events, canonical GB/s and graph-layers/s."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np
import torch

from gnm import core, synth
from gnm.arena import GraphArena


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--pool", type=int, default=256)
    ap.add_argument("--F", type=int, default=64)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--backward", action="store_true")
    ap.add_argument("--fused", action="store_true",
                    help="gnm_agg_fwd_bnrelu: previous layer's BatchNorm+ReLU+readout on the tile load (what a step runs)")
    ap.add_argument("--phase-a-only", action="store_true", help="y = null: tile load + barrier only")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    pool = synth.make_pool("dense_fc", args.pool)
    ar = GraphArena(dev)
    gids = np.array([ar.add(g) for g in pool], dtype=np.int64)
    rng = np.random.default_rng(0)
    batch = ar.batch_from_gids(gids[rng.integers(0, args.pool, args.batch)])
    N, F = batch.N, args.F
    E = int(pool[0].edge_mat.shape[1])
    x = torch.randn(N, F, device=dev)
    h = torch.randn(N, F, device=dev)
    y = torch.empty(N, F, device=dev)
    eps = torch.zeros(1, device=dev)
    spec = core.GinSpec(5, 2, True, "sum", "sum")
    part = torch.empty(core.lib.gnm_agg_num_partials(F, batch.n_max, batch.B), dtype=torch.float64, device=dev)

    sc, sh = torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev) * 0.3
    gf = torch.empty(batch.B, F, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        if args.fused:
            core.check(core.lib.gnm_agg_fwd_bnrelu(
                ar.rowptr.buf.data_ptr(), ar.col.buf.data_ptr(), batch.rp_off.data_ptr(), batch.col_off.data_ptr(),
                batch.node_off.data_ptr(), batch.B, batch.n_max, batch.nnz_max, x.data_ptr(), F, sc.data_ptr(),
                sh.data_ptr(), h.data_ptr(), F, gf.data_ptr(), F, 0, y.data_ptr(), F, F, eps.data_ptr(), 0, 0, st),
                "gnm_agg_fwd_bnrelu")
        elif args.backward:
            core._agg(batch, x, y, F, eps.data_ptr(), spec, True, hfwd=h, deps_partial=part)
        elif args.phase_a_only:
            core._agg(batch, x, None, F, eps.data_ptr(), spec, False)
        else:
            core._agg(batch, x, y, F, eps.data_ptr(), spec, False)

    for _ in range(5):
        run()
    torch.cuda.synchronize()
    evs = []
    for _ in range(args.iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in evs])
    byt = (4 * 400 * F * 2 + 4 * E + 4 * 401) * args.batch
    med = float(np.median(ms))
    print("agg %s F=%d B=%d: median %.1f us (min %.1f)  %.0f GB/s canonical (%.1f%% of 8 TB/s)  %.2f M graph-layers/s"
          % ("bwd" if args.backward else "fwd", F, args.batch, med * 1e3, ms.min() * 1e3, byt / med / 1e6,
             100 * byt / med / 1e6 / 8000, args.batch / med / 1e3))
    # correctness spot check vs a dense torch reference on one graph
    g0 = pool[int(rng.integers(0, 1))]
    b1 = ar.batch_from_gids(gids[:2])
    x1 = torch.randn(b1.N, F, device=dev)
    y1 = torch.empty_like(x1)
    core._agg(b1, x1, y1, F, eps.data_ptr(), spec, False)
    ref = []
    for k in range(2):
        em = pool[k].edge_mat.to(dev)
        A = torch.zeros(400, 400, device=dev, dtype=torch.float64)
        A.index_put_((em[0], em[1]), torch.ones(em.shape[1], device=dev, dtype=torch.float64), accumulate=True)
        xs = x1[400 * k:400 * (k + 1)].double()
        ref.append(A @ xs + xs)
    ref = torch.cat(ref)
    err = (y1.double() - ref).abs().max().item() / ref.abs().max().item()
    print("spot check vs dense fp64: rel err %.2e" % err)
    assert err < 1e-5


if __name__ == "__main__":
    main()
