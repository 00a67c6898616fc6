#!/usr/bin/env python3
"""In-kernel timeline of the matrix-core aggregation kernel (csrc/aggm.hip; tuning build with -DGNM_AGG16_TUNING):
    python tools/build_variant.py tuning -DGNM_AGG16_TUNING
    GNM_HIP_LIB=graph-neural-mapping_amd/lib/variants/tuning.so python tools/aggm_timeline.py [--mode plain|fused|bwdstats]
Lane 0 of every wave stores s_memtime at: 0 entry, 1 tile loads issued, 2 planes written (before the barrier), 3 past
the barrier with its bit rows loaded, 4 product done, 5 first row block's epilogue done, 6 phase B left.  The counter is
per XCD and not the shader clock: phases are reported as shares of the workgroup's lifetime, next to the HIP-event
duration of the launch."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np
import torch

from gnm import core, synth
from gnm.arena import GraphArena

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="plain", choices=["plain", "fused", "bwdstats"])
ap.add_argument("--batch", type=int, default=1024)
args = ap.parse_args()
lib = core.lib
lib.gnm_debug_set_aggm_stamps.argtypes = [C.c_void_p]
lib.gnm_debug_set_aggm_stamps.restype = None
dev = torch.device("cuda:0")
pool = synth.make_pool("dense_fc", args.batch)
ar = GraphArena(dev)
batch = ar.batch_from_gids(np.array(ar.add_many(pool), dtype=np.int64))
assert batch.dense
N, F, B = batch.N, 64, batch.B
x, h, y = torch.randn(N, F, device=dev), torch.randn(N, F, device=dev), torch.empty(N, F, device=dev)
eps = torch.zeros(1, device=dev)
part = torch.empty(lib.gnm_aggm_num_partials(F, B), dtype=torch.float64, device=dev)
sc, sh = torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev) * 0.3
mu, rs = torch.randn(F, device=dev) * 0.1, torch.rand(F, device=dev) + 0.5
gf, dpool, U = torch.empty(B, F, device=dev), torch.randn(B, F, device=dev), torch.randn(B, F, device=dev)
dsc1, s2sum = torch.randn(N, device=dev), torch.randn(B, device=dev)
inv_perm = torch.randperm(B, device=dev).to(torch.int32)
spart = torch.empty(B, 2, F, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
a = ar
head = (a.rowptr.buf.data_ptr(), a.col.buf.data_ptr())


def run():
    if args.mode == "plain":
        core.check(lib.gnm_aggm(*head, batch.rp_off.data_ptr(), batch.col_off.data_ptr(), a.bits.buf.data_ptr(),
                                batch.bits_off.data_ptr(), a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(),
                                batch.node_off.data_ptr(), B, batch.n_max, x.data_ptr(), F, y.data_ptr(), F, F,
                                eps.data_ptr(), 0, 0, 0, None, 0, None, st), "gnm_aggm")
    elif args.mode == "fused":
        core.check(lib.gnm_aggm_fwd_bnrelu(*head, batch.rp_off.data_ptr(), batch.col_off.data_ptr(),
                                           a.bits.buf.data_ptr(), batch.bits_off.data_ptr(), batch.node_off.data_ptr(), B,
                                           batch.n_max, x.data_ptr(), F, sc.data_ptr(), sh.data_ptr(), h.data_ptr(), F,
                                           gf.data_ptr(), F, 0, y.data_ptr(), F, F, eps.data_ptr(), 0, 0, st), "fused")
    else:
        core.check(lib.gnm_aggm_bwd_stats(*head, batch.t_rp_off.data_ptr(), batch.t_col_off.data_ptr(),
                                          a.bits.buf.data_ptr(), batch.t_bits_off.data_ptr(), a.rowptr.buf.data_ptr(),
                                          batch.rp_off.data_ptr(), batch.node_off.data_ptr(), B, batch.n_max, x.data_ptr(),
                                          F, y.data_ptr(), F, F, eps.data_ptr(), 0, 0, None, 0, part.data_ptr(),
                                          h.data_ptr(), F, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(),
                                          dpool.data_ptr(), F, 0, dsc1.data_ptr(), U.data_ptr(), F, inv_perm.data_ptr(),
                                          s2sum.data_ptr(), spart.data_ptr(), st), "bwdstats")


for _ in range(5):
    run()
WG = ((B + 7) // 8) * 8 * (F // 32)
stamps = torch.zeros(WG * 8 * 16, dtype=torch.int64, device=dev)
lib.gnm_debug_set_aggm_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize(); stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.gnm_debug_set_aggm_stamps(None)
ms = e0.elapsed_time(e1)
s = stamps.cpu().numpy().reshape(WG, 8, 16).astype(np.float64)
t0 = s[:, :, 8].min(1, keepdims=True)
life = (s[:, :, 6].max(1, keepdims=True) - t0)
print("%s: launch %.1f us (stamped build), %d workgroups of 8 waves; workgroup lifetime %0.f ticks (median)" %
      (args.mode, ms * 1e3, WG, np.median(life)))
# at 2 workgroups per CU the launch is WG / 512 workgroup lifetimes long
print("  launch / (workgroups per slot = %d / 512) = %.1f us per workgroup lifetime -> 1 tick ~ %.2f ns" %
      (WG, ms * 1e3 / (WG / 512.0), ms * 1e6 / (WG / 512.0) / np.median(life)))
names = ["issue tile loads", "wait for the tile + split into planes + LDS writes", "barrier (+ readout) + bit rows",
         "product (MFMA)", "epilogue, first row block", "epilogue, second row block"]
active = s[:, :, 4] > s[:, :, 3]                       # waves that own a row block
for k, nm in enumerate(names):
    d = (s[:, :, k + 1] - s[:, :, k]) / life
    w = active if k >= 3 else np.ones_like(active)
    print("    %-52s %5.1f %% of the workgroup's lifetime (waves with work)" % (nm, 100 * d[w].mean()))
print("    %-52s %5.1f %% / %5.1f %%" % ("  of which: entry -> graph descriptor read / -> bit rows requested",
                                       100 * ((s[:, :, 0] - s[:, :, 8]) / life).mean(), 100 * ((s[:, :, 7] - s[:, :, 0]) / life).mean()))
print("    %-52s %5.1f %%" % ("idle at the end (wave done, workgroup not)", 100 * ((s[:, :, 6].max(1, keepdims=True) - s[:, :, 6]) / life).mean()))
two = (s[:, :, 6] - s[:, :, 5]) > 0.02 * life
print("    waves with two row blocks: %.2f of 8; with none: %.2f" % (two.sum(1).mean(), (~active).sum(1).mean()))
