#!/usr/bin/env python3
"""Times gnm_maxpool_fwd / gnm_maxpool_bwd (neighbor_pooling_type == "max") on dense 400-node graphs, F = 64."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np
import torch

from gnm import synth
from gnm._cabi import check, lib
from gnm.maxnb import MaxNeighbours


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--F", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = "cuda:0"
    pool = []
    for s in range(8):                                   # 8 distinct graphs, repeated: the lists are Python objects
        g = synth.dense_fc_graph(s, n=400, f0=1)
        em = g.edge_mat.numpy()
        order = np.argsort(em[0], kind="stable")
        cnt = np.bincount(em[0], minlength=400)
        g.neighbors = [x.tolist() for x in np.split(em[1][order], np.cumsum(cnt)[:-1])]
        g.max_neighbor = int(cnt.max())
        pool.append(g)
    graphs = [pool[i % 8] for i in range(args.batch)]
    mb = MaxNeighbours(graphs, False, dev)
    N, F = mb.N, args.F
    h = torch.relu(torch.randn(N, F, device=dev))
    out, g = torch.empty(N, F, device=dev), torch.randn(N, F, device=dev)
    dh = torch.empty(N, F, device=dev)
    amax = torch.empty(N, F, dtype=torch.int32, device=dev)
    eps = torch.zeros(1, device=dev)
    nblk = lib.gnm_maxpool_colmin_blocks(N)
    wv, wi = torch.empty(nblk, F, device=dev), torch.empty(nblk, F, dtype=torch.int32, device=dev)
    dummy, amin = torch.empty(F, device=dev), torch.empty(F, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        check(lib.gnm_maxpool_colmin(h.data_ptr(), F, N, F, wv.data_ptr(), wi.data_ptr(), dummy.data_ptr(), amin.data_ptr(),
                                     st), "colmin")
        check(lib.gnm_maxpool_fwd(h.data_ptr(), F, mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), N, F, mb.max_deg, 0,
                                  eps.data_ptr(), dummy.data_ptr(), out.data_ptr(), F, amax.data_ptr(), st), "fwd")

    def bwd():
        check(lib.gnm_maxpool_bwd(g.data_ptr(), F, amax.data_ptr(), mb.t_off.data_ptr(), mb.t_col.data_ptr(), N, F,
                                  eps.data_ptr(), None, 0, None, dh.data_ptr(), F, st), "bwd")

    node_off = torch.arange(args.batch + 1, dtype=torch.int32, device=dev) * 400

    def fwd_t():
        check(lib.gnm_maxpool_colmin(h.data_ptr(), F, N, F, wv.data_ptr(), wi.data_ptr(), dummy.data_ptr(), amin.data_ptr(),
                                     st), "colmin")
        check(lib.gnm_maxpool_fwd_tiled(h.data_ptr(), F, mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), node_off.data_ptr(),
                                        args.batch, 400, F, mb.max_deg, 0, eps.data_ptr(), dummy.data_ptr(), out.data_ptr(), F,
                                        amax.data_ptr(), st), "fwd_tiled")

    def bwd_t():
        check(lib.gnm_maxpool_bwd_tiled(g.data_ptr(), F, amax.data_ptr(), mb.t_off.data_ptr(), mb.t_col.data_ptr(),
                                        node_off.data_ptr(), args.batch, 400, F, eps.data_ptr(), None, 0, None, dh.data_ptr(),
                                        F, st), "bwd_tiled")

    for name, fn in (("fwd (colmin + gather-max)", fwd), ("bwd", bwd), ("fwd tiled (colmin + LDS)", fwd_t),
                     ("bwd tiled", bwd_t)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / args.iters
        edges = int(mb.nb_off[-1])
        print("MAXPOOL %-26s B=%d N=%d F=%d entries=%d: %.1f us  (%.0f G candidate-elements/s, gathered %.2f TB/s from cache)"
              % (name, args.batch, N, F, edges, us, edges * F / us / 1e3, edges * F * 4 / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
