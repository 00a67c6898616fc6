#!/usr/bin/env python3
"""Transposed-role aggregation kernel (csrc/aggt.hip) against the matrix-core kernel it would replace (csrc/aggm.hip):
plain and fused-prologue forms, ragged graph sizes, on random inputs; then the timing of both at the headline batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-neural-mapping_amd"))
import numpy as np, torch
from gnm import synth, core, _cabi
from gnm.arena import GraphArena
lib = core.lib
dev = torch.device("cuda:0")
lib.gnm_aggt.restype = _cabi.SIGNATURES["gnm_aggm"][0]; lib.gnm_aggt.argtypes = _cabi.SIGNATURES["gnm_aggm"][1]
lib.gnm_aggt_fwd_bnrelu.restype = _cabi.SIGNATURES["gnm_aggm_fwd_bnrelu"][0]
lib.gnm_aggt_fwd_bnrelu.argtypes = _cabi.SIGNATURES["gnm_aggm_fwd_bnrelu"][1]
st = torch.cuda.current_stream().cuda_stream


def case(pool, F, tag):
    ar = GraphArena(dev)
    gids = np.array(ar.add_many(pool), dtype=np.int64)
    batch = ar.batch_from_gids(gids)
    N, B = batch.N, batch.B
    torch.manual_seed(N + F)
    x = torch.randn(N, F, device=dev) * 3
    eps = torch.full((1,), 0.25, device=dev)
    sc, sh = torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev) * 0.3
    out = {}
    for name, fn, ffn in (("m", lib.gnm_aggm, lib.gnm_aggm_fwd_bnrelu), ("t", lib.gnm_aggt, lib.gnm_aggt_fwd_bnrelu)):
        y = torch.full((N, F), float("nan"), device=dev)
        core.check(fn(ar.rowptr.buf.data_ptr(), ar.col.buf.data_ptr(), batch.rp_off.data_ptr(), batch.col_off.data_ptr(),
                      ar.bits.buf.data_ptr(), batch.bits_off.data_ptr(), ar.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(),
                      batch.node_off.data_ptr(), B, batch.n_max, x.data_ptr(), F, y.data_ptr(), F, F, eps.data_ptr(), 0, 0, 0,
                      None, 0, None, st), name + " plain")
        y2 = torch.full((N, F), float("nan"), device=dev)
        hh = torch.full((N, F), float("nan"), device=dev)
        gf = torch.full((B, F), float("nan"), device=dev)
        core.check(ffn(ar.rowptr.buf.data_ptr(), ar.col.buf.data_ptr(), batch.rp_off.data_ptr(), batch.col_off.data_ptr(),
                       ar.bits.buf.data_ptr(), batch.bits_off.data_ptr(), batch.node_off.data_ptr(), B, batch.n_max,
                       x.data_ptr(), F, sc.data_ptr(), sh.data_ptr(), hh.data_ptr(), F, gf.data_ptr(), F, 0, y2.data_ptr(), F, F,
                       eps.data_ptr(), 0, 0, st), name + " fused")
        torch.cuda.synchronize()
        out[name] = [t.cpu().numpy().astype(np.float64) for t in (y, y2, hh, gf)]
    for k, what in enumerate(("plain y", "fused y", "activation", "readout")):
        a, b = out["m"][k], out["t"][k]
        assert np.isfinite(b).all(), (tag, what, "non-finite")
        err = np.abs(a - b).max() / (np.abs(a).max() + 1e-30)
        print("%-22s F=%3d %-11s max-norm relative difference %.2e" % (tag, F, what, err))
        assert err < 2e-6, (tag, what, err)


rng = np.random.default_rng(0)
case(synth.make_pool("dense_fc", 24, n=400, f0=7), 64, "24 x 400 dense")
case(synth.make_pool("dense_fc", 5, n=37, f0=7) + synth.make_pool("dense_fc", 4, n=416, f0=7) + synth.make_pool("dense_fc", 3, n=129, f0=7), 64, "ragged 37/416/129")
case(synth.make_pool("dense_fc", 9, n=100, f0=7) + synth.make_pool("dense_fc", 3, n=17, f0=7), 64, "9 x 100 + 3 x 17")
print("aggt ok")
