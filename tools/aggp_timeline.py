#!/usr/bin/env python3
"""In-kernel timeline of the PERSISTENT matrix-core aggregation kernel (gnm_aggp_kernel; tuning build):
    python tools/build_variant.py tuning -DGNM_AGG16_TUNING
    GNM_HIP_LIB=graph-neural-mapping_amd/lib/variants/tuning.so python tools/aggp_timeline.py [--mode plain|fused|bwdstats]
Lane 0 of every wave stamps s_memtime per unit: 0 past the hand-over, 1 scalars + requests issued (product starts),
2 product done, 3 epilogue done, 4 bits requested + reductions written, 5 next item split, 6 past the next hand-over."""
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
lib.gnm_aggm_persistent_min_units(1)
dev = torch.device("cuda:0")
pool = synth.make_pool("dense_fc", args.batch)
ar = GraphArena(dev)
batch = ar.batch_from_gids(np.array(ar.add_many(pool), dtype=np.int64))
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
                                           batch.n_max, x.data_ptr(), F, sc.data_ptr(), sh.data_ptr(), None, F,
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
WG = 256
stamps = torch.zeros(WG * 16 * 8 * 8, dtype=torch.int64, device=dev)
lib.gnm_debug_set_aggm_stamps(stamps.data_ptr())
run(); torch.cuda.synchronize(); stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.gnm_debug_set_aggm_stamps(None)
ms = e0.elapsed_time(e1)
s = stamps.cpu().numpy().reshape(WG, 16, 8, 8).astype(np.float64)[:, :13]
units = int((s[0, 0, :, 0] > 0).sum())
life = s[:, :, units - 1, 6].max(1) - s[:, :, 0, 0].min(1)
tick_ns = ms * 1e6 / np.median(life)
print("%s: launch %.1f us (stamped build); %d units per workgroup; kernel body %.0f ticks (median) -> 1 tick ~ %.2f ns"
      % (args.mode, ms * 1e3, units, np.median(life), tick_ns))
names = ["hand-over -> product start (scalars, next item + operand requests)", "product (MFMA)", "epilogue",
         "bits request + reductions", "split of the next item", "wait at the hand-over"]
per_unit = (s[:, :, 1:units - 1, 6] - s[:, :, 1:units - 1, 0])
print("  unit time (steady state, units 1..%d): median %.0f ticks = %.2f us" % (units - 2, np.median(per_unit), np.median(per_unit) * tick_ns * 1e-3))
for k, nm in enumerate(names):
    d = s[:, :, 1:units - 1, k + 1] - s[:, :, 1:units - 1, k]
    print("    %-70s %6.0f ticks  %5.1f %%" % (nm, np.median(d), 100 * np.median(d) / np.median(per_unit)))
