#!/usr/bin/env python3
"""Micro-benchmark of the aggregation kernel alone (F = 64 by default) through the C-ABI: median launch time from
HIP events, canonical GB/s and graph-layers/s, for the launch forms a training step uses:

  plain     gnm_agg forward
  fused     gnm_agg_fwd_bnrelu (previous layer's BatchNorm+ReLU+readout on the tile load: what a step's layers 1.. run)
  bwd       gnm_agg backward with the d-eps dot product on the way in
  bwdstats  gnm_agg_bwd_stats (backward + the layer below's BatchNorm-backward pass 1: what a step's backward runs)
  phasea    y = null: tile load + barrier only
  mplain / mfused / mbwdstats   the same three launch forms on the matrix-core kernel (csrc/aggm.hip, bit adjacency)

  python tools/bench_agg.py --modes plain,fused,bwdstats [--cold]
--cold streams 1 GiB between launches so inputs come from HBM, not from the Infinity Cache."""
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
    ap.add_argument("--pool", type=int, default=1024)
    ap.add_argument("--F", type=int, default=64)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--modes", default="plain,fused,bwdstats")
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--keep-hidden", action="store_true",
                    help="fused forms: also write the activation h = relu(bn(z)) (the model does not, by default)")
    ap.add_argument("--density", type=float, default=None,
                    help="random symmetric graphs of --nodes nodes with this edge density instead of the benchmark's "
                         "dense-FC graphs (where does the gather beat the matrix-core kernel?)")
    ap.add_argument("--nodes", type=int, default=400)
    ap.add_argument("--knn", action="store_true",
                    help="BASELINE configs[3]: 1000-node kNN (k = 20) graphs; use with --F 128 --batch 256 --pool 256")
    ap.add_argument("--tag", default=os.environ.get("GNM_HIP_LIB", "product"))
    ap.add_argument("--check", action="store_true", help="spot-check the result against a dense fp64 product")
    ap.add_argument("--ab", default=None,
                    help="comma-separated extra builds of the library (paths) timed INTERLEAVED with the loaded one: "
                         "launch i of every build before launch i + 1 of any -- clock drift and neighbours hit all alike")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.knn:
        pool = synth.make_pool("knn", args.pool, n=1000, f0=7)
    elif args.density is None:
        pool = synth.make_pool("dense_fc", args.pool)
    else:
        rng0 = np.random.default_rng(1)
        pool = []
        for _ in range(min(args.pool, 64)):
            src, dst = np.nonzero(np.triu(rng0.random((args.nodes, args.nodes)) < args.density, 1))
            pool.append(synth.SynthGraph(args.nodes, np.stack([src, dst], 1), np.zeros((args.nodes, 7), np.float32), 0))
        pool = [pool[k % len(pool)] for k in range(args.pool)]
    ar = GraphArena(dev)
    gids = np.array(ar.add_many(pool), dtype=np.int64)
    rng = np.random.default_rng(0)
    sel = rng.permutation(args.pool)[:args.batch] if args.pool >= args.batch else rng.integers(0, args.pool, args.batch)
    batch = ar.batch_from_gids(gids[sel])
    N, F, B = batch.N, args.F, batch.B
    E = int(np.mean([int(g.edge_mat.shape[1]) for g in pool[:64]]))
    x = torch.randn(N, F, device=dev)
    h = torch.randn(N, F, device=dev)
    y = torch.empty(N, F, device=dev)
    eps = torch.zeros(1, device=dev)
    spec = core.GinSpec(5, 2, True, "sum", "sum")
    part = torch.empty(max(core.lib.gnm_agg_num_partials(F, batch.n_max, B), core.lib.gnm_aggm_num_partials(F, B)),
                       dtype=torch.float64, device=dev)
    sc, sh = torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev) * 0.3
    mu, rs = torch.randn(F, device=dev) * 0.1, torch.rand(F, device=dev) + 0.5
    gf = torch.empty(B, F, device=dev)
    dpool = torch.randn(B, F, device=dev)
    U = torch.randn(B, F, device=dev)
    dsc1 = torch.randn(N, device=dev)
    inv_perm = torch.randperm(B, device=dev).to(torch.int32)
    s2sum = torch.randn(B, device=dev)
    spart = torch.empty(B, 2, F, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    a = ar
    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev) if args.cold else None

    import ctypes as C
    from gnm import _cabi
    libs = [(os.path.basename(str(args.tag)), core.lib)]
    for path in (args.ab.split(",") if args.ab else []):
        L = C.CDLL(path)
        for name, (res, at) in _cabi.SIGNATURES.items():
            if hasattr(L, name):
                getattr(L, name).restype = res
                getattr(L, name).argtypes = at
        libs.append((os.path.basename(path), L))

    def run(mode, lib=None):
        lib = lib or core.lib
        if mode in ("bwd", "phasea", "plain"):
            bw = mode == "bwd"
            core.check(lib.gnm_agg(a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.rp_off.data_ptr(),
                                   batch.col_off.data_ptr(), a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(),
                                   batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max, x.data_ptr(), F,
                                   None if mode == "phasea" else y.data_ptr(), 0 if mode == "phasea" else F, F,
                                   eps.data_ptr(), 0, 0, int(bw), h.data_ptr() if bw else None, F if bw else 0,
                                   part.data_ptr() if bw else None, st), "gnm_agg")
            return
        if mode == "mplain":
            core.check(lib.gnm_aggm(a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.rp_off.data_ptr(),
                                    batch.col_off.data_ptr(), a.bits.buf.data_ptr(), batch.bits_off.data_ptr(),
                                    a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(), batch.node_off.data_ptr(), B,
                                    batch.n_max, x.data_ptr(), F, y.data_ptr(), F, F, eps.data_ptr(), 0, 0, 0, None, 0,
                                    None, st), "gnm_aggm")
            return
        if mode == "mfused":
            core.check(lib.gnm_aggm_fwd_bnrelu(
                a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.rp_off.data_ptr(), batch.col_off.data_ptr(),
                a.bits.buf.data_ptr(), batch.bits_off.data_ptr(), batch.node_off.data_ptr(), B, batch.n_max,
                x.data_ptr(), F, sc.data_ptr(), sh.data_ptr(), h.data_ptr() if args.keep_hidden else None, F, gf.data_ptr(),
                F, 0, y.data_ptr(), F, F, eps.data_ptr(), 0, 0, st), "gnm_aggm_fwd_bnrelu")
            return
        if mode == "mbwdstats":
            core.check(lib.gnm_aggm_bwd_stats(
                a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(), batch.t_col_off.data_ptr(),
                a.bits.buf.data_ptr(), batch.t_bits_off.data_ptr(), a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(),
                batch.node_off.data_ptr(), B, batch.n_max, x.data_ptr(), F, y.data_ptr(), F, F, eps.data_ptr(), 0, 0,
                None, 0, part.data_ptr(), h.data_ptr(), F, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(),
                dpool.data_ptr(), F, 0, dsc1.data_ptr(), U.data_ptr(), F, inv_perm.data_ptr(), s2sum.data_ptr(),
                spart.data_ptr(), st), "gnm_aggm_bwd_stats")
            return
        if mode == "fused":
            core.check(lib.gnm_agg_fwd_bnrelu(
                a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.rp_off.data_ptr(), batch.col_off.data_ptr(),
                batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max, x.data_ptr(), F, sc.data_ptr(),
                sh.data_ptr(), h.data_ptr() if args.keep_hidden else None, F, gf.data_ptr(), F, 0, y.data_ptr(), F, F,
                eps.data_ptr(), 0, 0, st), "gnm_agg_fwd_bnrelu")
        elif mode == "bwd":
            core._agg(batch, x, y, F, eps.data_ptr(), spec, True, hfwd=h, deps_partial=part)
        elif mode == "bwdstats":
            core.check(lib.gnm_agg_bwd_stats(
                a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(), batch.t_col_off.data_ptr(),
                a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(), batch.node_off.data_ptr(), B, batch.n_max,
                batch.nnz_max, x.data_ptr(), F, y.data_ptr(), F, F, eps.data_ptr(), 0, 0, None, 0, part.data_ptr(),
                h.data_ptr(), F, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), dpool.data_ptr(), F, 0,
                dsc1.data_ptr(), U.data_ptr(), F, inv_perm.data_ptr(), s2sum.data_ptr(), spart.data_ptr(), st),
                "gnm_agg_bwd_stats")
        elif mode == "phasea":
            core._agg(batch, x, None, F, eps.data_ptr(), spec, False)
        else:
            core._agg(batch, x, y, F, eps.data_ptr(), spec, False)

    n_nodes = len(pool[0].g)
    byt = (4 * n_nodes * F * 2 + 4 * E + 4 * (n_nodes + 1)) * args.batch
    for mode in args.modes.split(","):
        for _, L in libs:
            for _ in range(5):
                run(mode, L)
        torch.cuda.synchronize()
        evs = {name: [] for name, _ in libs}
        for _ in range(args.iters):
            for name, L in libs:
                if flush is not None:
                    flush.add_(1.0)
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(); run(mode, L); eb.record()
                evs[name].append((ea, eb))
        torch.cuda.synchronize()
        base = None
        for name, _ in libs:
            ms = np.array([ea.elapsed_time(eb) for ea, eb in evs[name]])
            med = float(np.median(ms))
            rel = "" if base is None else "   paired vs %s: %+.1f us (median of differences)" % (
                libs[0][0], 1e3 * float(np.median(ms - base)))
            if base is None:
                base = ms
            print("AGG %-28s %-8s F=%d B=%d%s: median %.1f us (min %.1f)  %.0f GB/s canonical = %.3f of 8 TB/s  "
                  "%.2f M graph-layers/s%s" % (name, mode, F, args.batch, " cold" if args.cold else "", med * 1e3,
                                              ms.min() * 1e3, byt / med / 1e6, byt / med / 1e6 / 8000,
                                              args.batch / med / 1e3, rel), flush=True)
    if args.check:
        b1 = ar.batch_from_gids(gids[:2])
        x1 = torch.randn(b1.N, F, device=dev)
        y1 = torch.empty_like(x1)
        core._agg(b1, x1, y1, F, eps.data_ptr(), spec, False)
        ref = []
        for k in range(2):
            em = pool[k].edge_mat.to(dev)
            nk = int(b1.node_off_host[k + 1] - b1.node_off_host[k])
            assert int(em.max()) < nk
            A = torch.zeros(nk, nk, device=dev, dtype=torch.float64)
            A.index_put_((em[0], em[1]), torch.ones(em.shape[1], device=dev, dtype=torch.float64), accumulate=True)
            xs = x1[int(b1.node_off_host[k]):int(b1.node_off_host[k + 1])].double()
            ref.append(A @ xs + xs)
        ref = torch.cat(ref)
        err = (y1.double() - ref).abs().max().item() / ref.abs().max().item()
        print("spot check vs dense fp64: rel err %.2e" % err)
        assert err < 1e-5


if __name__ == "__main__":
    main()
