"""Forward/backward orchestration of the GIN hot path over the C-ABI kernels.

One torch.autograd.Function (GinInfoMaxFn) covers everything GIN_InfoMaxReg.forward
computes after batch assembly (/root/reference models/graphcnn.py:208-251): the L GIN
layers (neighbour aggregation -> MLP -> BatchNorm -> ReLU), the per-layer graph readout and
classifier, and the Infomax discriminator scores.  Its backward is hand-derived (the same
derivation as oracle/gin_oracle.py, which the tests check it against) and launches the
HIP kernels of csrc/ directly; torch only provides device memory, the stream, and a few
tiny [B, .] dense ops (classifier Linear(H, C), U = sigmoid(g_f) W^T).

Every N-sized array op runs in libgnm_hip.so.  There is no CPU or eager-PyTorch fallback
for the sum/average path: on a non-GPU tensor the calls raise.
"""
import ctypes as C
import os

import numpy as np

import torch
import torch.nn.functional as F

from ._cabi import GnmError, check, lib, ptr

BN_EPS = 1e-5       # nn.BatchNorm1d defaults (mlp.py:38, graphcnn.py:51)
BN_MOMENTUM = 0.1
# the dW / db partials of the fused Linear backwards are reduced by ONE launch at the end of the backward
# (GNM_NO_DEFER_REDUCE=1: one launch after each Linear, for A/B timing)
DEFER_WGRAD_REDUCE = os.environ.get("GNM_NO_DEFER_REDUCE") is None


_STREAM = None      # raw hipStream_t of torch's current stream, fetched once per forward / backward


def _stream():
    return _STREAM if _STREAM is not None else torch.cuda.current_stream().cuda_stream


def launch_device(*tensors):
    """The one CUDA device all of `tensors` (None entries skipped) live on; GnmError otherwise.  The kernels
    take raw pointers, so nothing below this check would notice a CPU tensor or a second device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise GnmError("the GIN hot path runs on the GPU only (libgnm_hip.so); got a %s tensor" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise GnmError("tensors on different devices (%s and %s): one model replica lives on one GPU"
                           % (dev, t.device))
    if dev is None:
        raise GnmError("no device tensor to launch on")
    return dev


class _stream_scope:
    """Makes `device` (the device of the tensors being worked on, NOT whatever torch.cuda.current_device() happens
    to be) current for the duration of a forward or backward, and resolves its current stream once:
    torch.cuda.current_stream() costs ~10 us of Python per call and there are ~110 launches per step.  Without
    the device switch a model built on cuda:1 in a process whose current device is 0 would launch on device 0's
    stream with device-1 pointers (autograd's backward thread sets the device, a custom Function.forward does not)."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        global _STREAM
        self.prev = _STREAM
        self.guard = torch.cuda.device(self.device)
        self.guard.__enter__()
        _STREAM = torch.cuda.current_stream(self.device).cuda_stream

    def __exit__(self, *exc):
        global _STREAM
        _STREAM = self.prev
        self.guard.__exit__(*exc)
        return False


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels are
    launched on (torch's current stream).  bench.py turns it on for the timed region to
    compute the roofline figures; it is off (None) otherwise."""

    def __init__(self, prefixes=None):
        self.records = []          # (name, meta, start_event, end_event)
        self.prefixes = tuple(prefixes) if prefixes else None   # only time launches whose tag starts with one

    def summary(self):
        """name -> (count, mean ms, meta); call after a synchronize.  Launches recorded under one name with
        DIFFERENT meta (e.g. the fused-prologue aggregation and the plain one) are different kernels: they are
        kept apart under "name|key=value,..." of the differing keys, never averaged together."""
        groups = {}
        for name, meta, a, b in self.records:
            key = (name, tuple(sorted(meta.items())))
            c, t, m = groups.get(key, (0, 0.0, meta))
            groups[key] = (c + 1, t + a.elapsed_time(b), m)
        by_name = {}
        for (name, _), v in groups.items():
            by_name.setdefault(name, []).append(v)
        out = {}
        for name, vs in by_name.items():
            if len(vs) == 1:
                c, t, m = vs[0]
                out[name] = (c, t / c, m)
                continue
            common = set.intersection(*[set(m.items()) for _, _, m in vs])
            for c, t, m in vs:
                diff = ",".join("%s=%s" % kv for kv in sorted(set(m.items()) - common))
                out["%s|%s" % (name, diff)] = (c, t / c, m)
        return out


TIMER = None


def _new_event():
    """HIP event on the launch stream (a seam the CPU tests replace with a fake clock)."""
    return torch.cuda.Event(enable_timing=True)


class _timed:
    def __init__(self, name, **meta):
        self.name, self.meta = name, meta

    def __enter__(self):
        self.on = TIMER is not None and (TIMER.prefixes is None or self.name.startswith(TIMER.prefixes))
        if self.on:
            self.a = _new_event()
            self.b = _new_event()
            self.a.record()

        return self

    def cancel(self):
        """Nothing was launched inside the block (the entry point declined the shape): record no interval."""
        self.on = False

    def __exit__(self, *exc):
        if self.on and TIMER is not None and exc[0] is None:
            self.b.record()
            TIMER.records.append((self.name, self.meta, self.a, self.b))
        return False


class GinSpec:
    """Static description of the model: which parameter tensor is which."""

    def __init__(self, num_layers, num_mlp_layers, learn_eps, graph_pooling_type, neighbor_pooling_type):
        self.L, self.m = num_layers, num_mlp_layers
        self.learn_eps = bool(learn_eps)
        self.g_avg = graph_pooling_type == "average"
        self.n_avg = neighbor_pooling_type == "average"
        # "max" (graphcnn.py:137-143): csrc/maxpool.hip over batch.maxnb (gnm/maxnb.py) in place of the aggregation
        # kernels; the BatchNorm + ReLU of the layer below then runs as its own kernel
        self.n_max = neighbor_pooling_type == "max"
        # Optional gradient sink {parameter name: tensor}: when set (gnm.parallel.DataParallelGIN
        # points it at views of its flat gradient buffer) the backward kernels write every parameter
        # gradient straight into these tensors (OVERWRITING them) and autograd gets None, instead of
        # fresh tensors that AccumulateGrad then adds into .grad with ~50 tiny kernels per step.
        self.grad_sink = None
        # Optional cross-rank BatchNorm (SURVEY.md 8(e) "sync_bn"): an object with all_reduce(tensor) (SUM over
        # the data-parallel group) and global_count(n).  When set, train-mode BatchNorm normalises with the
        # statistics of the UNION batch, so W ranks reproduce one process on the whole batch.  Eager launches only.
        self.sync_bn = None
        # True: every layer's activation h_l = relu(bn(z_l)) is written to memory as an array (test hooks that read
        # them).  Default: layers whose BatchNorm + ReLU ride on the next aggregation's tile load exist only as ZAct
        # (z, scale, shift) -- the discriminator re-forms them in its kernels; 105 MB per layer less HBM traffic.
        self.keep_hidden = False


def _dense(batch, F_, spec=None):
    """does this batch take the matrix-core aggregation (csrc/aggm.hip)?  The arena decides per batch (dense graphs
    with a bit adjacency, GraphArena.batch_from_gids); the kernel wants whole 32-column blocks, or one partial block
    (the input layer's F0 < 32, plain form only: the fused / d-eps forms decline and the gather runs).
    One mode keeps a dense batch on the CSR gather: neighbour "average" + learn_eps with a node that has no
    neighbours.  That node's row is 0/0 = NaN in the reference (graphcnn.py:157-158) and stays confined to the rows
    that gather it; a product multiplies it by the zero bits of every other row of its graph (0 x NaN = NaN)."""
    if spec is not None and spec.n_avg and spec.learn_eps and getattr(batch, "iso", False):
        return False
    return bool(getattr(batch, "dense", False)) and (F_ % 32 == 0 or F_ < 32)


def agg_partials_capacity(batch, F_):
    """doubles a d-eps partial buffer must hold for this batch, whichever aggregation kernel runs"""
    k = int(lib.gnm_agg_num_partials(F_, batch.n_max, batch.B))
    if _dense(batch, F_):           # (an upper bound: whichever kernel the mode selects)
        k = max(k, int(lib.gnm_aggm_num_partials(F_, batch.B)))
    return k


def _agg(batch, x, y, F_, eps_ptr, spec, backward, hfwd=None, deps_partial=None):
    """y = None: only the d-eps partials are produced (no gather).  Returns the number of d-eps partials written."""
    a = batch.arena
    if backward:
        rp_off, col_off = batch.t_rp_off, batch.t_col_off
    else:
        rp_off, col_off = batch.rp_off, batch.col_off
    tag = "agg_%s_F%d%s" % ("bwd" if backward else "fwd", F_, "_dot" if y is None else "")
    if _dense(batch, F_, spec):
        bits_off = batch.t_bits_off if backward else batch.bits_off
        with _timed(tag, F=F_, B=batch.B, N=batch.N, mfma=1) as tm:
            rc = lib.gnm_aggm(a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), rp_off.data_ptr(), col_off.data_ptr(),
                              a.bits.buf.data_ptr(), bits_off.data_ptr(), a.rowptr.buf.data_ptr(),
                              batch.rp_off.data_ptr(), batch.node_off.data_ptr(), batch.B, batch.n_max, x.data_ptr(),
                              x.stride(0), ptr(y), y.stride(0) if y is not None else 0, F_, eps_ptr, int(spec.n_avg),
                              int(not spec.learn_eps), int(backward), ptr(hfwd),
                              hfwd.stride(0) if hfwd is not None else 0, ptr(deps_partial), _stream())
            if rc != 0:
                tm.cancel()
        if rc == 0:
            return int(lib.gnm_aggm_num_partials(F_, batch.B))
        if rc != -2:
            check(rc, "gnm_aggm")
    with _timed(tag, F=F_, B=batch.B, N=batch.N):
        check(lib.gnm_agg(a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), rp_off.data_ptr(), col_off.data_ptr(),
                          a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(), batch.node_off.data_ptr(), batch.B,
                          batch.n_max, batch.nnz_max, x.data_ptr(), x.stride(0), ptr(y),
                          y.stride(0) if y is not None else 0, F_,
                          eps_ptr, int(spec.n_avg), int(not spec.learn_eps), int(backward), ptr(hfwd),
                          hfwd.stride(0) if hfwd is not None else 0, ptr(deps_partial), _stream()), "gnm_agg")
    return int(lib.gnm_agg_num_partials(F_, batch.n_max, batch.B))


def _max_fwd(batch, h, pooled, F_, eps_ptr):
    """pooled = max over the neighbour rows of h [+ (1 + eps) h] (graphcnn.py:137-143, 149-151/161, 173-175).
    Returns (amax, amin): the selected row of every element and torch.min's row per column -- where the backward
    sends the gradient."""
    mb = getattr(batch, "maxnb", None)
    if mb is None or mb.N != batch.N:
        raise GnmError("neighbor_pooling_type='max' needs the graphs' neighbour lists: call forward(batch_graph)")
    N = batch.N
    dev = h.device
    dummy = amin = None
    if mb.need_dummy:                                   # some row is padded: dummy = torch.min(h, dim=0)[0]  (:140)
        nblk = int(lib.gnm_maxpool_colmin_blocks(N))
        wv = torch.empty((nblk, F_), dtype=torch.float32, device=dev)
        wi = torch.empty((nblk, F_), dtype=torch.int32, device=dev)
        dummy = torch.empty(F_, dtype=torch.float32, device=dev)
        amin = torch.empty(F_, dtype=torch.int32, device=dev)
        check(lib.gnm_maxpool_colmin(h.data_ptr(), h.stride(0), N, F_, wv.data_ptr(), wi.data_ptr(), dummy.data_ptr(),
                                     amin.data_ptr(), _stream()), "gnm_maxpool_colmin")
    if mb.max_deg == 0 and not mb.self_last and N > 0:
        raise IndexError("max(): Expected reduction dim 1 to have non-zero size.")     # what torch.max raises (:142)
    amax = torch.empty((N, F_), dtype=torch.int32, device=dev)
    with _timed("maxpool_fwd_F%d" % F_, F=F_, B=batch.B, N=N):
        # one workgroup per graph with its rows in LDS when the shape allows, else rows gathered from L2: same bits
        rc = lib.gnm_maxpool_fwd_tiled(h.data_ptr(), h.stride(0), mb.nb_off.data_ptr(), mb.nb_col.data_ptr(),
                                       batch.node_off.data_ptr(), batch.B, batch.n_max, F_, mb.max_deg,
                                       int(mb.self_last), eps_ptr, ptr(dummy), pooled.data_ptr(), pooled.stride(0),
                                       amax.data_ptr(), _stream())
        if rc == -2:
            rc = lib.gnm_maxpool_fwd(h.data_ptr(), h.stride(0), mb.nb_off.data_ptr(), mb.nb_col.data_ptr(), N, F_,
                                     mb.max_deg, int(mb.self_last), eps_ptr, ptr(dummy), pooled.data_ptr(),
                                     pooled.stride(0), amax.data_ptr(), _stream())
        check(rc, "gnm_maxpool_fwd")
    return amax, amin


def _max_bwd(batch, dpooled, dh, F_, eps_ptr, aux, hfwd, deps_partial):
    """d h from d pooled through the selection recorded by _max_fwd; d eps partials as a flat dot product.
    Returns the number of d-eps partials written."""
    mb = batch.maxnb
    amax, amin = aux
    with _timed("maxpool_bwd_F%d" % F_, F=F_, B=batch.B, N=batch.N):
        rc = lib.gnm_maxpool_bwd_tiled(dpooled.data_ptr(), dpooled.stride(0), amax.data_ptr(), mb.t_off.data_ptr(),
                                       mb.t_col.data_ptr(), batch.node_off.data_ptr(), batch.B, batch.n_max, F_,
                                       eps_ptr, ptr(mb.iso_rows), mb.n_iso, ptr(amin), dh.data_ptr(), dh.stride(0),
                                       _stream())
        if rc == -2:
            rc = lib.gnm_maxpool_bwd(dpooled.data_ptr(), dpooled.stride(0), amax.data_ptr(), mb.t_off.data_ptr(),
                                     mb.t_col.data_ptr(), batch.N, F_, eps_ptr, ptr(mb.iso_rows), mb.n_iso, ptr(amin),
                                     dh.data_ptr(), dh.stride(0), _stream())
        check(rc, "gnm_maxpool_bwd")
    if deps_partial is None:
        return 0
    check(lib.gnm_rowdot_partials(dpooled.data_ptr(), dpooled.stride(0), hfwd.data_ptr(), hfwd.stride(0), batch.N, F_,
                                  deps_partial.data_ptr(), _stream()), "gnm_rowdot_partials")
    return int(lib.gnm_rowdot_num_partials())


def _linear(x, W, w_kmajor, bias, z, N, K, H, pro, stats):
    with _timed("lin_%s_K%d_H%d" % ("dgrad" if w_kmajor else "fwd", K, H), N=N, K=K, H=H):
        check(lib.gnm_linear_fwd(x.data_ptr(), x.stride(0), W.data_ptr(), W.stride(0), int(w_kmajor), ptr(bias),
                                 z.data_ptr(), z.stride(0), N, K, H, ptr(pro[0]) if pro else None,
                                 ptr(pro[1]) if pro else None, 1 if pro else 0, ptr(stats), _stream()),
              "gnm_linear_fwd")


def _linear_wide(x, W, w_kmajor, bias, z, N, K, H, pro, stats):
    """H > 128 (only dX of a wide first layer): column windows of 128."""
    if H <= 128:
        return _linear(x, W, w_kmajor, bias, z, N, K, H, pro, stats)
    assert stats is None
    for h0 in range(0, H, 128):
        hw = min(128, H - h0)
        Wv = W[:, h0:h0 + hw] if w_kmajor else W[h0:h0 + hw]
        _linear(x, Wv, w_kmajor, bias[h0:h0 + hw] if bias is not None else None, z[:, h0:h0 + hw], N, K, hw, pro, None)


# the K = H = 64 Linear backward recomputes the Linear's output from its input instead of reading it
# (gnm_linear_bwd_fused_rz); GNM_NO_RZ=1 keeps the form that reads the stored output (A/B and tests)
RZ_BACKWARD = os.environ.get("GNM_NO_RZ", "0") != "1"


# GNM_NO_MASKED_DGRAD=1: the H = 128 Linears' dX without the fused ReLU mask / BatchNorm sums (A/B timing, tests)
MASKED_DGRAD = os.environ.get("GNM_NO_MASKED_DGRAD", "0") != "1"

# the three [B, L*H]-sized products of the Infomax tail on the hand-written kernel (csrc/sgemm.hip); GNM_NO_SGEMM=1
# keeps torch.mm (hipBLASLt) for A/B timing
SMALL_GEMM = os.environ.get("GNM_NO_SGEMM", "0") != "1"


def _small_gemm(A, a_cols, B, b_cols, out, M, N, K):
    """out[M,N] = A' B' (csrc/sgemm.hip: gnm_small_gemm); False when the kernel declines and the caller uses torch."""
    if not SMALL_GEMM or A.stride(1) != 1 or B.stride(1) != 1 or out.stride(1) != 1:
        return False
    rc = lib.gnm_small_gemm(A.data_ptr(), A.stride(0), int(a_cols), B.data_ptr(), B.stride(0), int(b_cols), out.data_ptr(),
                            out.stride(0), M, N, K, _stream())
    if rc == -2:
        return False
    check(rc, "gnm_small_gemm")
    return True


class _LinSave:
    __slots__ = ("x_in", "pro", "z", "scale", "shift", "mean", "rstd", "K", "H", "Ng")


def encoder_forward(spec, batch, X, P, training, update_running, P0=None):
    """The L GIN layers + readout.  P: dict of parameter/buffer tensors keyed by the
    reference's state_dict names.  P0: the arena's cached parameter-independent part of layer 0's
    aggregation (GraphArena.features_and_agg0) or None.  Returns (hidden list, g_f [B, L*H], saved)."""
    dev = X.device
    N, B = batch.N, batch.B
    L, m = spec.L, spec.m
    H = P["batch_norms.0.weight"].shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    g_f = torch.empty((B, L * H), **f32)
    hidden, saved = [], []
    h = X
    sync = spec.sync_bn if training else None
    Ng = sync.global_count(N, dev) if sync is not None else N        # rows of the union batch
    a = batch.arena
    pending = None      # (z, scale, shift, hout, gslice) of the previous layer: its BatchNorm+ReLU+readout not yet run

    def readout(z, scale, shift, hout, gslice):
        check(lib.gnm_bn_relu_readout(z.data_ptr(), z.stride(0), scale.data_ptr(), shift.data_ptr(),
                                      ptr(hout), hout.stride(0) if hout is not None else 0,
                                      batch.node_off.data_ptr(), B, H, 1,
                                      gslice.data_ptr(), g_f.stride(0), int(spec.g_avg), _stream()),
              "gnm_bn_relu_readout")                                          # graphcnn.py:163-166, 228-229

    for l in range(L):
        F_l = h.shape[1]
        eps_ptr = P["eps"].data_ptr() + 4 * l if spec.learn_eps else None
        aux = None          # max pooling: what its backward needs
        if l == 0 and P0 is not None:
            # A X [/deg] comes from the arena's cache; only the (1 + eps_0) X self term depends on a parameter
            pooled = torch.addcmul(P0, h, P["eps"][0:1] + 1.0) if spec.learn_eps else P0
        else:
            pooled = torch.empty((N, F_l), **f32)
            fused = False
            if pending is not None:
                # the previous layer's BatchNorm + ReLU + readout ride on this aggregation's tile load; unless the
                # caller wants the arrays (spec.keep_hidden) the activation itself is not written
                z, scale, shift, hout, gslice = pending
                hout_ptr, hout_ld = (hout.data_ptr(), hout.stride(0)) if hout is not None else (None, 0)
                rc = -2
                if _dense(batch, F_l, spec) and not spec.n_max:
                    with _timed("agg_fwd_F%d" % F_l, F=F_l, B=B, N=N, fused_bnrelu=1, mfma=1) as tm:
                        rc = lib.gnm_aggm_fwd_bnrelu(
                            a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.rp_off.data_ptr(),
                            batch.col_off.data_ptr(), a.bits.buf.data_ptr(), batch.bits_off.data_ptr(),
                            batch.node_off.data_ptr(), B, batch.n_max, z.data_ptr(), z.stride(0), scale.data_ptr(),
                            shift.data_ptr(), hout_ptr, hout_ld, gslice.data_ptr(), g_f.stride(0),
                            int(spec.g_avg), pooled.data_ptr(), pooled.stride(0), F_l, eps_ptr, int(spec.n_avg),
                            int(not spec.learn_eps), _stream())
                        if rc != 0:
                            tm.cancel()
                if rc == -2 and not spec.n_max:
                    with _timed("agg_fwd_F%d" % F_l, F=F_l, B=B, N=N, fused_bnrelu=1) as tm:
                        rc = lib.gnm_agg_fwd_bnrelu(
                            a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.rp_off.data_ptr(),
                            batch.col_off.data_ptr(), batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max,
                            z.data_ptr(), z.stride(0), scale.data_ptr(), shift.data_ptr(), hout_ptr,
                            hout_ld, gslice.data_ptr(), g_f.stride(0), int(spec.g_avg), pooled.data_ptr(),
                            pooled.stride(0), F_l, eps_ptr, int(spec.n_avg), int(not spec.learn_eps), _stream())
                        if rc != 0:
                            tm.cancel()       # declined (or failed): no launch happened, the fallback below times itself
                if rc == -2:
                    if hout is None:                 # declined: the unfused pair of kernels needs the array after all
                        hout = torch.empty((N, z.shape[1]), **f32)
                        hidden[-1] = hout
                        h = hout
                    readout(z, scale, shift, hout, gslice)
                else:
                    check(rc, "gnm_agg_fwd_bnrelu")
                    fused = True
                pending = None
            if spec.n_max:
                aux = _max_fwd(batch, hidden_tensor(h), pooled, F_l, eps_ptr)
            elif not fused:
                _agg(batch, hidden_tensor(h), pooled, F_l, eps_ptr, spec, backward=False)    # graphcnn.py:154-161 / 178-182
        x_in, pro, lins = pooled, None, []
        for k in range(m):                                                   # mlp.py:40-49
            if m == 1:
                W, bias = P[f"mlps.{l}.linear.weight"], P[f"mlps.{l}.linear.bias"]
            else:
                W, bias = P[f"mlps.{l}.linears.{k}.weight"], P[f"mlps.{l}.linears.{k}.bias"]
            bn = f"batch_norms.{l}" if k == m - 1 else f"mlps.{l}.batch_norms.{k}"
            Hk, K = W.shape
            z = torch.empty((N, Hk), **f32)
            grid = lib.gnm_linear_grid(N)
            stats = torch.empty((grid, 2, Hk), dtype=torch.float64, device=dev) if training else None
            _linear(x_in, W, 0, bias, z, N, K, Hk, pro, stats)
            sv = _LinSave()
            sv.x_in, sv.pro, sv.z, sv.K, sv.H = x_in, pro, z, K, Hk
            sv.scale, sv.shift, sv.mean, sv.rstd = (torch.empty(Hk, **f32) for _ in range(4))
            sv.Ng = Ng
            if sync is not None:
                stats = stats.sum(0, keepdim=True)            # [1,2,Hk] column sum / sum of squares (fp64)
                sync.all_reduce(stats)
                grid = 1
            check(lib.gnm_bn_finalize(ptr(stats), grid, Hk, Ng, P[bn + ".weight"].data_ptr(),
                                      P[bn + ".bias"].data_ptr(), P[bn + ".running_mean"].data_ptr(),
                                      P[bn + ".running_var"].data_ptr(),
                                      P[bn + ".num_batches_tracked"].data_ptr(), BN_MOMENTUM, BN_EPS,
                                      int(training), int(update_running), sv.scale.data_ptr(), sv.shift.data_ptr(),
                                      sv.mean.data_ptr(), sv.rstd.data_ptr(), _stream()), "gnm_bn_finalize")
            lins.append(sv)
            x_in, pro = z, (sv.scale, sv.shift)
        gslice = g_f[:, l * H:(l + 1) * H]
        if l < L - 1:
            # deferred into the next layer's aggregation, which (on its fused path) does not write the activation
            hout = torch.empty((N, H), **f32) if getattr(spec, "keep_hidden", False) else None
            pending = (x_in, pro[0], pro[1], hout, gslice)
            hnew = hout if hout is not None else ZAct(x_in, pro[0], pro[1])
        else:
            # the top layer: only its readout is needed now; the discriminator re-forms the activation like the others
            hout = torch.empty((N, H), **f32) if getattr(spec, "keep_hidden", False) else None
            readout(x_in, pro[0], pro[1], hout, gslice)
            hnew = hout if hout is not None else ZAct(x_in, pro[0], pro[1])
        saved.append((h, pooled, lins, aux))
        hidden.append(hnew)
        h = hnew
    return hidden, g_f, saved


class DiscUnit:
    """Hand-over between the discriminator scores' forward, the loss and the backward (round 3).

    With the reference's loss on d_logit -- BCEWithLogits against ones / zeros, main.py:32-37 -- the backward's
    per-graph reductions (dU, s2sum, dsum of gnm_disc_score_bwd) are k x quantities that depend on the FORWARD values
    only, k being the loss's scalar factor times the upstream gradient.  gnm_disc_score_fwd_unit leaves them in `unit`
    from the hidden rows it holds in registers anyway; a loss that knows it has that form (gnm.train.infomax_loss with
    its default targets) records k and the gradient tensor it handed to autograd; GinInfoMaxFn.backward then scales
    `unit` instead of reading the five hidden layers a second time (gnm_disc_du_kernel: 539 MB, ~90 us at B = 1024).
    Any other loss (main.py's torch losses, explicit d_labels, a d_logit that also feeds something else so that
    autograd SUMS gradients into a new tensor) leaves k unset or the pointer different, and the backward runs
    gnm_disc_score_bwd exactly as before."""
    __slots__ = ("unit", "inv_perm", "k", "kscale", "dD_ptr", "dD_version")

    def __init__(self):
        self.unit = self.inv_perm = self.k = self.dD_ptr = self.dD_version = None
        self.kscale = 1.0


# GNM_NO_DISC_UNIT=1: never produce the by-products (A/B timing against gnm_disc_score_bwd's pass)
DISC_UNIT = os.environ.get("GNM_NO_DISC_UNIT") is None


class ZAct:
    """A layer output that is not in memory: h = relu(z * scale + shift) (graphcnn.py:163-166) with z the pre-BatchNorm
    output of the layer's last Linear and (scale, shift) its folded BatchNorm.  Its BatchNorm + ReLU ran on the tile
    load of the next layer's aggregation, which did not write it (105 MB per layer at the
    headline shape); the only other consumer, the discriminator, re-forms it in its kernels from the same three
    tensors.  tensor() materialises it for the rare paths that want the array (unfused fallbacks, test hooks)."""
    __slots__ = ("z", "scale", "shift", "_h")

    def __init__(self, z, scale, shift):
        self.z, self.scale, self.shift, self._h = z, scale, shift, None

    @property
    def shape(self):
        return self.z.shape

    def tensor(self):
        if self._h is None:
            self._h = torch.relu(torch.addcmul(self.shift, self.z, self.scale))
        return self._h


def hidden_tensor(h):
    """the [N, H] array of a layer output, whichever way encoder_forward holds it"""
    return h.tensor() if isinstance(h, ZAct) else h


def _hptr_array(hidden):
    arr = (C.c_void_p * len(hidden))(*[t.data_ptr() for t in hidden])
    return arr


def _hidden_ptr_arrays(hidden):
    """(pointers, scale pointers, shift pointers, leading dimension) for gnm_disc_score_fwd / _bwd: a ZAct layer is
    passed as its z with the BatchNorm vectors, a materialised one as itself (NULL vectors)"""
    n = len(hidden)
    hp, sp, tp = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    ld = None
    for l, h in enumerate(hidden):
        if isinstance(h, ZAct):
            hp[l], sp[l], tp[l] = h.z.data_ptr(), h.scale.data_ptr(), h.shift.data_ptr()
            stride = h.z.stride(0)
        else:
            hp[l], sp[l], tp[l] = h.data_ptr(), None, None
            stride = h.stride(0)
        if ld is not None and stride != ld:
            raise GnmError("hidden layers with different leading dimensions (%d, %d)" % (ld, stride))
        ld = stride
    return hp, sp, tp, ld


_ptr_array = _hptr_array


# GNM_NO_EVAL_FUSED=1: eval-mode forwards through the layer-by-layer kernels only (A/B timing, tests)
EVAL_FUSED = os.environ.get("GNM_NO_EVAL_FUSED") is None


def eval_fused_ok(spec, batch, X, P, mode=True):
    """can this eval-mode forward run as the one-launch encoder (csrc/evalfwd.hip; mode True) / as one launch per layer
    with a workgroup per 32-row block (csrc/evallayer.hip; mode "layers")?"""
    if not EVAL_FUSED or spec.n_max or spec.sync_bn is not None or spec.keep_hidden:
        return False
    layers = mode == "layers"
    if not getattr(batch, "has_bits", False) or batch.B < 1 or batch.n_max > (416 if layers else int(lib.gnm_eval_max_nodes())):
        return False
    if spec.n_avg and spec.learn_eps and getattr(batch, "iso", False):
        return False            # the 0/0 row of an isolated node must stay confined to its row (see _dense)
    H = P["batch_norms.0.weight"].shape[0]
    C_ = P["linears_prediction.0.weight"].shape[0]
    if layers:
        return H in (32, 64, 128) and 1 <= spec.m <= 3 and spec.L <= 16 and X.shape[1] <= 128 and C_ <= 256 and X.is_cuda
    return H == 64 and 1 <= spec.m <= 3 and spec.L <= 16 and X.shape[1] <= 64 and C_ <= 64 and X.is_cuda


def check_permutation(perm, B):
    """The Infomax shuffle (graphcnn.py:199: np.random.permutation(B)) as a host int array, validated: the kernels
    scatter its inverse and index rows with it unchecked, so anything that is not a permutation of 0 .. B-1 must stop
    here (round 3 clamped on the device instead and would have trained on wrong gradients)."""
    p = np.asarray(perm)
    if p.ndim != 1 or p.shape[0] != B or p.dtype.kind not in "iu":
        raise GnmError("perm must be %d integers (np.random.permutation(B), graphcnn.py:199); got shape %s dtype %s"
                       % (B, p.shape, p.dtype))
    if B and (int(p.min()) < 0 or int(p.max()) >= B or not bool((np.bincount(p, minlength=B) == 1).all())):
        raise GnmError("perm is not a permutation of 0 .. %d" % (B - 1))
    return p


def perm_to_device(perm, B, dev, out=None):
    """perm -> int32 device tensor (or into the static buffer `out`).  Host permutations are validated and go through
    pinned memory (a pageable copy would stall the host behind everything queued); a CUDA tensor is taken as it is --
    it can only come from a buffer that was filled through this function."""
    if torch.is_tensor(perm) and perm.is_cuda:
        t = perm.to(torch.int32)
        if out is not None and out.data_ptr() != t.data_ptr():
            out.copy_(t, non_blocking=True)
        return t if out is None else out
    host = torch.as_tensor(check_permutation(perm.cpu().numpy() if torch.is_tensor(perm) else perm, B).astype(np.int32))
    if dev.type == "cuda":
        host = host.pin_memory()
    if out is not None:
        out.copy_(host, non_blocking=True)
        return out
    return host.to(dev, non_blocking=True)


def eval_forward_fused(spec, batch, perm, P, X, want_disc, mode=True):
    """GIN_InfoMaxReg.forward in eval() mode (graphcnn.py:194-251 with BatchNorm on its running statistics and dropout
    off) as ONE encoder launch (gnm_eval_encoder: layers + readout + classifier, a workgroup per graph) plus, for the
    Infomax scores, U = sigmoid(g_f) W^T and the score kernel.  No autograd graph: callers use it under no_grad only
    (models/graphcnn.py).  Returns (c_logit, d_logit, g_f) like GinInfoMaxFn."""
    dev = launch_device(X, *[t for t in P.values() if torch.is_tensor(t)][:1])
    L, m = spec.L, spec.m
    N, B = batch.N, batch.B
    H = P["batch_norms.0.weight"].shape[0]
    Cn = P["linears_prediction.0.weight"].shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    X = X.contiguous()
    a = batch.arena
    words = []
    for l in range(L):
        for k in range(m):
            wn = f"mlps.{l}.linear" if m == 1 else f"mlps.{l}.linears.{k}"
            bn = f"batch_norms.{l}" if k == m - 1 else f"mlps.{l}.batch_norms.{k}"
            W = P[wn + ".weight"]
            if W.stride(1) != 1:
                raise GnmError("Linear weights must be row-contiguous")
            words += [W.data_ptr(), P[wn + ".bias"].data_ptr(), P[bn + ".weight"].data_ptr(), P[bn + ".bias"].data_ptr(),
                      P[bn + ".running_mean"].data_ptr(), P[bn + ".running_var"].data_ptr(), W.stride(0)]
    for l in range(L):
        wp, bp = P[f"linears_prediction.{l}.weight"], P[f"linears_prediction.{l}.bias"]
        if not wp.is_contiguous() or not bp.is_contiguous():
            raise GnmError("classifier parameters must be contiguous")
        words += [wp.data_ptr(), bp.data_ptr()]
    assert len(words) == int(lib.gnm_eval_table_words(L, m))
    # the table of parameter addresses lives on the device; rebuilt only when a parameter moved (a few hundred bytes,
    # one pinned asynchronous copy -- capturable)
    key = tuple(words)
    cached = getattr(spec, "_eval_table", None)
    if cached is None or cached[0] != key or cached[1].device != dev:
        host = torch.tensor(words, dtype=torch.int64).pin_memory()
        cached = spec._eval_table = (key, host.to(dev, non_blocking=True), host)
    table = cached[1]
    hidden_all = torch.empty((L, N, H), **f32)
    hidden = [hidden_all[l] for l in range(L)]
    g_f = torch.empty((B, L * H), **f32)
    c = torch.empty_like(g_f) if want_disc else None
    c_logit = torch.empty((B, Cn), **f32)
    if mode == "layers":
        scratch = torch.empty(int(lib.gnm_eval_layers_scratch_floats(B, batch.n_max, H, L)), **f32)
        with _stream_scope(dev):
            check(lib.gnm_eval_layers(
                a.bits.buf.data_ptr(), batch.bits_off.data_ptr(), batch.node_off.data_ptr(), a.rowptr.buf.data_ptr(),
                batch.rp_off.data_ptr(), B, batch.n_max, X.data_ptr(), X.stride(0), X.shape[1], H, L, m, Cn,
                int(spec.n_avg), int(not spec.learn_eps), int(spec.g_avg), BN_EPS, table.data_ptr(),
                P["eps"].data_ptr() if spec.learn_eps else None, hidden_all.data_ptr(), hidden_all.stride(0), H,
                scratch.data_ptr(), g_f.data_ptr(), g_f.stride(0), ptr(c), c_logit.data_ptr(), c_logit.stride(0),
                _stream()), "gnm_eval_layers")
    else:
      s0, s1 = torch.empty((N, H), **f32), torch.empty((N, H), **f32)
      with _stream_scope(dev):
        check(lib.gnm_eval_encoder(
            a.bits.buf.data_ptr(), batch.bits_off.data_ptr(), batch.node_off.data_ptr(), a.rowptr.buf.data_ptr(),
            batch.rp_off.data_ptr(), B, batch.n_max, X.data_ptr(), X.stride(0), X.shape[1], H, L, m, Cn,
            int(spec.n_avg), int(not spec.learn_eps), int(spec.g_avg), BN_EPS, table.data_ptr(),
            P["eps"].data_ptr() if spec.learn_eps else None, hidden_all.data_ptr(), hidden_all.stride(0), H,
            s0.data_ptr(), s1.data_ptr(), H, g_f.data_ptr(), g_f.stride(0), ptr(c), c_logit.data_ptr(),
            c_logit.stride(0), _stream()), "gnm_eval_encoder")
    with _stream_scope(dev):
        d_logit = torch.zeros((0, 1), **f32)
        if want_disc:
            if not batch.equal_n:
                raise RuntimeError("Discriminator expands each graph summary N//B times (discriminator.py:24): "
                                   "all graphs of a batch must have the same number of nodes")
            Wd = P["disc.f_k.weight"][0]
            U = torch.empty((B, Wd.shape[0]), **f32)
            if not _small_gemm(c, 0, Wd, 0, U, B, Wd.shape[0], Wd.shape[1]):
                U = c @ Wd.t()                                                # U[g] = W c_g
            perm_rows = perm_to_device(perm, B, dev)
            d_logit = torch.empty((2 * N, 1), **f32)
            hp_, sp_, tp_, ldh_ = _hidden_ptr_arrays(hidden)
            check(lib.gnm_disc_score_fwd(hp_, sp_, tp_, ldh_, L, H, U.data_ptr(), U.stride(0), perm_rows.data_ptr(),
                                         P["disc.f_k.bias"].data_ptr(), batch.node_off.data_ptr(), N, B,
                                         d_logit.data_ptr(), _stream()), "gnm_disc_score_fwd")
    return c_logit, d_logit, g_f


class GinInfoMaxFn(torch.autograd.Function):
    """(P0, X, *params) -> (c_logit [B,C], d_logit [2N,1], g_f [B,L*H])."""

    @staticmethod
    def forward(ctx, spec, batch, perm, names, buffers, training, dropout_p, want_disc, P0, X, *tensors):
        dev = launch_device(X, P0, *tensors)
        adev = batch.arena.device
        if adev.type != "cuda" or (adev.index is not None and adev.index != dev.index):
            raise GnmError("the batch's graph arena lives on %s, the model on %s" % (adev, dev))
        ctx.set_materialize_grads(False)
        with _stream_scope(dev):
            return GinInfoMaxFn._forward(ctx, spec, batch, perm, names, buffers, training, dropout_p, want_disc, P0,
                                         X, tensors)

    @staticmethod
    def _forward(ctx, spec, batch, perm, names, buffers, training, dropout_p, want_disc, P0, X, tensors):
        P = dict(zip(names, tensors))
        P.update(buffers)
        L = spec.L
        N, B = batch.N, batch.B
        X = X.contiguous()
        hidden, g_f, saved = encoder_forward(spec, batch, X, P, training, update_running=training, P0=P0)
        H = hidden[0].shape[1]
        # classifier head (graphcnn.py:224-231) and sigmoid(g_f) (:239): one launch (csrc/head.hip)
        wps = [P[f"linears_prediction.{l}.weight"] for l in range(L)]
        bps = [P[f"linears_prediction.{l}.bias"] for l in range(L)]
        Cn = wps[0].shape[0]
        masks = None
        if training and dropout_p > 0:
            # (F.dropout of a cached tensor of ones: the fill it would otherwise need is a launch per step)
            ones = getattr(spec, "_ones_mask", None)
            if ones is None or ones.shape != (L, B, Cn) or ones.device != X.device:
                ones = spec._ones_mask = torch.ones((L, B, Cn), dtype=torch.float32, device=X.device)
            masks = F.dropout(ones, dropout_p, True)                                                      # :230
        c_logit = torch.empty((B, Cn), dtype=torch.float32, device=X.device)
        c = torch.empty_like(g_f) if want_disc else None
        fused_head = all(w.is_contiguous() for w in wps) and all(b_.is_contiguous() for b_ in bps)
        if fused_head:
            rc = lib.gnm_head_fwd(g_f.data_ptr(), g_f.stride(0), B, L, H, Cn, _ptr_array(wps), _ptr_array(bps),
                                  ptr(masks), c_logit.data_ptr(), c_logit.stride(0), ptr(c),
                                  c.stride(0) if c is not None else 0, _stream())
            if rc == -2:
                fused_head = False
            else:
                check(rc, "gnm_head_fwd")
        Wp = None
        if not fused_head:          # shapes outside the head kernel (C > 256 classes): batched matrix products
            Wp = torch.stack(wps)                                                            # [L,C,H]
            G3 = g_f.view(B, L, H).transpose(0, 1)                                           # [L,B,H] view
            lg = torch.baddbmm(torch.stack(bps).unsqueeze(1), G3, Wp.transpose(1, 2))        # [L,B,C]
            if masks is not None:
                lg = lg * masks
            c_logit = lg.sum(0)
            if want_disc:
                c = torch.sigmoid(g_f)
        d_logit = None
        U = perm_rows = None
        disc_unit = None
        if want_disc:
            if not batch.equal_n:
                raise RuntimeError("Discriminator expands each graph summary N//B times (discriminator.py:24): "
                                   "all graphs of a batch must have the same number of nodes")
            Wd = P["disc.f_k.weight"][0]
            LH = Wd.shape[1]
            U = torch.empty((B, Wd.shape[0]), dtype=torch.float32, device=X.device)
            if not _small_gemm(c, 0, Wd, 0, U, B, Wd.shape[0], LH):
                U = c @ Wd.t()                                                # U[g] = W c_g
            # row index = perm[g] (:198-201,242): validated on the host, pinned staging + async copy (a device tensor --
            # a captured step's static buffer -- passes as it is: graph-capture safe)
            perm_rows = perm_to_device(perm, B, X.device)
            d_logit = torch.empty((2 * N, 1), dtype=torch.float32, device=X.device)
            hp_, sp_, tp_, ldh_ = _hidden_ptr_arrays(hidden)
            rc = -2
            tm_ = _timed("disc_score", N=N, L=L, H=H)
            tm_.__enter__()
            if isinstance(want_disc, DiscUnit) and training:
                # also leave the backward's per-graph reductions (up to the loss's scalar factor): see DiscUnit
                ldunit = (L * H + 2 + 3) & ~3
                unit = torch.empty((B, ldunit), dtype=torch.float32, device=X.device)
                inv_perm = torch.empty(B, dtype=torch.int32, device=X.device)
                rc = lib.gnm_disc_score_fwd_unit(hp_, sp_, tp_, ldh_, L, H, U.data_ptr(), U.stride(0),
                                                 perm_rows.data_ptr(), P["disc.f_k.bias"].data_ptr(),
                                                 batch.node_off.data_ptr(), N, B, d_logit.data_ptr(), unit.data_ptr(),
                                                 ldunit, inv_perm.data_ptr(), _stream())
                if rc == 0:
                    want_disc.unit, want_disc.inv_perm = unit, inv_perm
                    disc_unit = want_disc
                elif rc != -2:
                    check(rc, "gnm_disc_score_fwd_unit")
            if rc == -2:
                check(lib.gnm_disc_score_fwd(hp_, sp_, tp_, ldh_, L, H, U.data_ptr(), U.stride(0),
                                             perm_rows.data_ptr(), P["disc.f_k.bias"].data_ptr(),
                                             batch.node_off.data_ptr(), N, B, d_logit.data_ptr(), _stream()),
                      "gnm_disc_score_fwd")
            tm_.__exit__(None, None, None)
        ctx.spec, ctx.batch, ctx.names, ctx.P = spec, batch, names, P
        ctx.hidden, ctx.saved, ctx.g_f, ctx.masks, ctx.Wp = hidden, saved, g_f, masks, Wp
        ctx.fused_head, ctx.wps = fused_head, wps
        ctx.c, ctx.U, ctx.perm_rows, ctx.perm = c, U, perm_rows, perm
        ctx.training, ctx.X = training, X
        ctx.disc_unit = disc_unit
        ctx.mark_non_differentiable(g_f)
        if d_logit is None:
            d_logit = torch.zeros((0, 1), dtype=torch.float32, device=X.device)
        return c_logit, d_logit, g_f

    @staticmethod
    def backward(ctx, dC, dD, _dgf):
        with _stream_scope(ctx.g_f.device):
            return GinInfoMaxFn._backward(ctx, dC, dD)

    @staticmethod
    def _backward(ctx, dC, dD):
        spec, batch, P = ctx.spec, ctx.batch, ctx.P
        L, m = spec.L, spec.m
        N, B = batch.N, batch.B
        hidden, saved, g_f = ctx.hidden, ctx.saved, ctx.g_f
        H = hidden[0].shape[1]
        dev = g_f.device
        f32 = dict(dtype=torch.float32, device=dev)
        st = _stream()
        grads = {}
        need_dx = ctx.needs_input_grad[9]
        sink = spec.grad_sink

        def out_like(name, ref):
            """tensor the kernel should write gradient `name` into"""
            return sink[name] if sink is not None else torch.empty_like(ref)

        def put(name, value):
            """record a gradient computed by a torch op"""
            if sink is not None:
                sink[name].copy_(value.reshape(sink[name].shape))
            else:
                grads[name] = value

        # ---- discriminator (discriminator.py:28-36) --------------------------------
        dsc1 = U = inv_perm = s2sum = None
        dg_f = T = None
        if dD is not None and ctx.U is not None:
            dD = dD.contiguous().view(-1)
            U, c = ctx.U, ctx.c
            dU = torch.empty_like(U)
            s2sum = torch.empty(B, **f32)
            dsum = torch.empty(B, **f32)
            dbias = None
            hold = ctx.disc_unit
            if (hold is not None and hold.k is not None and hold.dD_ptr == dD.data_ptr()
                    and hold.dD_version == dD._version):      # (same tensor AND untouched since the loss wrote it)
                # dD = k (sigmoid(d_logit) - target) came straight from the loss that recorded k: the reductions are k
                # times what the forward left (DiscUnit) -- no second pass over the hidden layers
                inv_perm = hold.inv_perm
                # (the Bilinear bias gradient -- the total of dsum -- comes out of the same launch)
                dbias = sink["disc.f_k.bias"] if sink is not None else torch.empty(1, **f32)
                check(lib.gnm_disc_unit_scale(hold.unit.data_ptr(), hold.unit.stride(0), L * H, hold.k.data_ptr(),
                                              float(getattr(hold, "kscale", 1.0)), B,
                                              dU.data_ptr(), dU.stride(0), s2sum.data_ptr(), dsum.data_ptr(),
                                              dbias.data_ptr(), st),
                      "gnm_disc_unit_scale")
            else:
                inv_perm = torch.empty(B, dtype=torch.int32, device=dev)      # inverse permutation, on the device
                hp_, sp_, tp_, ldh_ = _hidden_ptr_arrays(hidden)
                with _timed("disc_du", N=N, L=L, H=H):
                    check(lib.gnm_disc_score_bwd(hp_, sp_, tp_, ldh_, L, H, dD.data_ptr(),
                                                 ctx.perm_rows.data_ptr(), batch.node_off.data_ptr(), N, B,
                                                 dU.data_ptr(), dU.stride(0), s2sum.data_ptr(), dsum.data_ptr(),
                                                 inv_perm.data_ptr(), st), "gnm_disc_score_bwd")
            if hold is not None:
                hold.k = hold.dD_ptr = hold.dD_version = None
            Wd = P["disc.f_k.weight"][0]
            # (the three [B, L*H]-sized products of the tail: csrc/sgemm.hip -- one workgroup per 32 x 32 output tile, the
            #  contraction split over its four waves, split-precision bf16 products.  A first hand-written version on the
            #  fp32 matrix instruction measured 48 us per product against hipBLASLt's 9-12 us.)
            Bc = dU.shape[0]
            dWd = sink["disc.f_k.weight"][0] if sink is not None else torch.empty_like(Wd)
            if not _small_gemm(dU, 1, c, 1, dWd, dU.shape[1], c.shape[1], Bc):          # dWd = dU^T c
                torch.mm(dU.t(), c, out=dWd)
            if sink is not None:
                if dbias is None:
                    torch.sum(dsum, 0, keepdim=True, out=sink["disc.f_k.bias"])
            else:
                grads["disc.f_k.weight"] = dWd.unsqueeze(0)
                grads["disc.f_k.bias"] = dbias if dbias is not None else dsum.sum().reshape(1)
            T = torch.empty((Bc, Wd.shape[1]), **f32)
            if not _small_gemm(dU, 0, Wd, 1, T, Bc, Wd.shape[1], Wd.shape[0]):            # d loss / d sigmoid(g_f) = dU Wd
                T = dU @ Wd
            dsc1 = dD                                                         # first N entries = d sc_1

        # ---- classifier head: d g_f (classifier + sigmoid paths) and the classifier gradients ------------
        dph = [None] * L
        if ctx.fused_head and (dC is not None or T is not None):
            wps = ctx.wps
            Cn = wps[0].shape[0]
            dCc = dC.contiguous() if dC is not None else torch.zeros((B, Cn), **f32)
            dws = [out_like(f"linears_prediction.{l}.weight", wps[l]) for l in range(L)]
            dbs = [out_like(f"linears_prediction.{l}.bias", wps[l][:, 0]) for l in range(L)]
            dph_all = torch.empty((B, L * H), **f32)
            check(lib.gnm_head_bwd(dCc.data_ptr(), dCc.stride(0), ptr(ctx.masks), g_f.data_ptr(), g_f.stride(0),
                                   ptr(ctx.c), ctx.c.stride(0) if ctx.c is not None else 0, ptr(T),
                                   T.stride(0) if T is not None else 0, B, L, H, Cn, _ptr_array(wps),
                                   _ptr_array(dws), _ptr_array(dbs), dph_all.data_ptr(), dph_all.stride(0), st),
                  "gnm_head_bwd")
            if sink is None:
                for l in range(L):
                    grads[f"linears_prediction.{l}.weight"] = dws[l]
                    grads[f"linears_prediction.{l}.bias"] = dbs[l]
            dph = [dph_all[:, l * H:(l + 1) * H] for l in range(L)]
        elif not ctx.fused_head:
            if T is not None:
                dg_f = T * ctx.c * (1 - ctx.c)                                # sigmoid backward
            dph_all = None
            G3 = g_f.view(B, L, H).transpose(0, 1)                                           # [L,B,H]
            if dC is not None:
                dlg = dC.unsqueeze(0).expand(L, -1, -1)
                if ctx.masks is not None:
                    dlg = dlg * ctx.masks
                dWp = torch.bmm(dlg.transpose(1, 2), G3)                                     # [L,C,H]
                dbp = dlg.sum(1)                                                             # [L,C]
                for l in range(L):
                    put(f"linears_prediction.{l}.weight", dWp[l])
                    put(f"linears_prediction.{l}.bias", dbp[l])
                if dg_f is not None:
                    dph_all = torch.baddbmm(dg_f.view(B, L, H).transpose(0, 1), dlg, ctx.Wp)   # [L,B,H] contiguous
                else:
                    dph_all = torch.bmm(dlg, ctx.Wp)
            elif dg_f is not None:
                dph_all = dg_f.view(B, L, H).transpose(0, 1).contiguous()
            dph = [dph_all[l] if dph_all is not None else None for l in range(L)]

        deps = eps_parts = None
        eps_counts = [0] * L
        if spec.learn_eps:
            deps = sink["eps"] if sink is not None else torch.empty(L, **f32)
            # fp64 partials of d eps[l] from the L aggregation backwards, summed by ONE launch at the end
            eps_stride = max([agg_partials_capacity(batch, sv_[0].shape[1]) for sv_ in saved] +
                             [int(lib.gnm_rowdot_num_partials())])
            eps_parts = torch.empty((L, eps_stride), dtype=torch.float64, device=dev)
        dH_next = None
        dX = None
        pre_outer = None
        wjobs = []          # (workspace, dW, db, H, K) of the fused Linear backwards whose partial reduction is deferred
        for l in reversed(range(L)):
            h_in, pooled, lins, aux = saved[l]       # h_in: this layer's input (X, an array, or a ZAct)
            F_l = h_in.shape[1]
            incoming = dH_next                       # grad wrt this layer's output from the layer above
            # (G, partial, nblk) when the producer of this gradient already applied the ReLU mask and reduced
            # the BatchNorm-backward sums: the aggregation backward of the layer above (outer BatchNorm) or
            # the fused backward of the Linear above (inner BatchNorms)
            pre_stats, pre_outer = pre_outer, None
            for k in reversed(range(m)):
                sv = lins[k]
                Hk, K = sv.H, sv.K
                last = k == m - 1
                bn = f"batch_norms.{l}" if last else f"mlps.{l}.batch_norms.{k}"
                dp = dph[l] if last else None
                use_disc = last and dsc1 is not None
                Ul = U[:, l * H:(l + 1) * H] if use_disc else None
                if pre_stats is not None:
                    G, part, nblk = pre_stats
                    pre_stats = None
                else:
                  G = torch.empty((N, Hk), **f32)
                  part = torch.empty((B, 2, Hk), dtype=torch.float64, device=dev)
                  nblk = B
                  check(lib.gnm_bn_relu_bwd_stats(
                      ptr(incoming), incoming.stride(0) if incoming is not None else 0,
                      ptr(dp), dp.stride(0) if dp is not None else 0, int(spec.g_avg),
                      ptr(dsc1) if use_disc else None, ptr(Ul), U.stride(0) if use_disc else 0,
                      ptr(inv_perm) if use_disc else None, ptr(s2sum) if use_disc else None,
                      sv.z.data_ptr(), sv.z.stride(0), sv.scale.data_ptr(), sv.shift.data_ptr(), sv.mean.data_ptr(),
                      sv.rstd.data_ptr(), 1, G.data_ptr(), G.stride(0), batch.node_off.data_ptr(), B, Hk,
                      part.data_ptr(), st), "gnm_bn_relu_bwd_stats")
                dgamma, dbeta = out_like(bn + ".weight", sv.scale), out_like(bn + ".bias", sv.scale)
                cA, m1, m2 = (torch.empty(Hk, **f32) for _ in range(3))
                check(lib.gnm_bn_bwd_finalize(part.data_ptr(), nblk, Hk, N, P[bn + ".weight"].data_ptr(),
                                              sv.rstd.data_ptr(), int(ctx.training), dgamma.data_ptr(),
                                              dbeta.data_ptr(), cA.data_ptr(), m1.data_ptr(), m2.data_ptr(), st),
                      "gnm_bn_bwd_finalize")
                if spec.sync_bn is not None and ctx.training:
                    # d gamma / d beta stay LOCAL sums (the gradient all-reduce averages them, as for every other
                    # parameter); the two means inside dZ = cA (G - m1 - xhat m2) are over the union batch
                    red = part.view(-1, 2, Hk)[:nblk].sum(0, keepdim=True)
                    spec.sync_bn.all_reduce(red)
                    scratch = torch.empty((2, Hk), **f32)
                    check(lib.gnm_bn_bwd_finalize(red.data_ptr(), 1, Hk, sv.Ng, P[bn + ".weight"].data_ptr(),
                                                  sv.rstd.data_ptr(), 1, scratch[0].data_ptr(),
                                                  scratch[1].data_ptr(), cA.data_ptr(), m1.data_ptr(),
                                                  m2.data_ptr(), st), "gnm_bn_bwd_finalize")
                if sink is None:
                    grads[bn + ".weight"], grads[bn + ".bias"] = dgamma, dbeta
                wname = f"mlps.{l}.linear" if m == 1 else f"mlps.{l}.linears.{k}"
                W = P[wname + ".weight"]
                dW = out_like(wname + ".weight", W)
                db = out_like(wname + ".bias", sv.scale)
                # dX of this Linear: always for inner Linears; for the first one only when the
                # aggregation backward below has a consumer (a lower layer, dX, or d eps[l])
                need_dA = k > 0 or l > 0 or need_dx or spec.learn_eps
                dA = torch.empty((N, K), **f32) if need_dA else None
                # one fused pass (BatchNorm-backward apply + dX + dW + db) when the shape is eligible
                ws = torch.empty(int(lib.gnm_linear_bwd_workspace_floats(N, Hk, K)), **f32)
                # if a BatchNorm+ReLU feeds this Linear (k > 0), let the same pass mask dX with that ReLU and
                # reduce that BatchNorm's backward sums (replaces its gnm_bn_relu_bwd_stats launch)
                lo = lins[k - 1] if k > 0 else None
                lo_part = None
                if lo is not None and need_dA:
                    lo_part = torch.empty((lib.gnm_linear_bwd_grid(N), 2, K), dtype=torch.float64, device=dev)
                with _timed("linbwd_K%d_H%d" % (K, Hk), N=N, K=K, H=Hk) as tm:
                    lo_args = (lo.z.data_ptr() if lo_part is not None else None,
                               lo.z.stride(0) if lo_part is not None else 0,
                               lo.scale.data_ptr() if lo_part is not None else None,
                               lo.shift.data_ptr() if lo_part is not None else None,
                               lo.mean.data_ptr() if lo_part is not None else None,
                               lo.rstd.data_ptr() if lo_part is not None else None, ptr(lo_part), st)
                    rc = -2
                    if RZ_BACKWARD and Hk == 64 and ((K == 64 and need_dA) or (K <= 16 and not sv.pro and lo_part is None)):
                        # sv.z = Linear(sv.x_in) as gnm_linear_fwd left it: the pass recomputes it instead of reading it
                        rc = lib.gnm_linear_bwd_fused_rz(
                            G.data_ptr(), G.stride(0), P[wname + ".bias"].data_ptr(), sv.mean.data_ptr(),
                            sv.rstd.data_ptr(), cA.data_ptr(), m1.data_ptr(), m2.data_ptr(), sv.x_in.data_ptr(),
                            sv.x_in.stride(0), ptr(sv.pro[0]) if sv.pro else None, ptr(sv.pro[1]) if sv.pro else None,
                            1 if sv.pro else 0, W.data_ptr(), W.stride(0), ptr(dA), dA.stride(0) if need_dA else 0,
                            None if DEFER_WGRAD_REDUCE else dW.data_ptr(), dW.stride(0), db.data_ptr(), ws.data_ptr(),
                            N, K, Hk, *lo_args)
                    if rc == -2:
                      rc = lib.gnm_linear_bwd_fused(
                        G.data_ptr(), G.stride(0), sv.z.data_ptr(), sv.z.stride(0), sv.mean.data_ptr(),
                        sv.rstd.data_ptr(), cA.data_ptr(), m1.data_ptr(), m2.data_ptr(), sv.x_in.data_ptr(),
                        sv.x_in.stride(0), ptr(sv.pro[0]) if sv.pro else None, ptr(sv.pro[1]) if sv.pro else None,
                        1 if sv.pro else 0, W.data_ptr(), W.stride(0), ptr(dA), dA.stride(0) if need_dA else 0,
                        None if DEFER_WGRAD_REDUCE else dW.data_ptr(), dW.stride(0), db.data_ptr(), ws.data_ptr(), N, K, Hk,
                        *lo_args)
                    if rc != 0:
                        tm.cancel()
                if rc == 0 and DEFER_WGRAD_REDUCE:
                    wjobs.append((ws, dW, db, Hk, K))       # its dW / db partials: one reduction launch after the loop
                if rc == 0 and lo_part is not None:
                    pre_stats = (dA, lo_part, lo_part.shape[0])
                if rc == -2:        # GNM_ERR_UNSUPPORTED: generic three-kernel path
                    check(lib.gnm_bn_bwd_apply(G.data_ptr(), G.stride(0), sv.z.data_ptr(), sv.z.stride(0),
                                               sv.mean.data_ptr(), sv.rstd.data_ptr(), cA.data_ptr(), m1.data_ptr(),
                                               m2.data_ptr(), G.data_ptr(), G.stride(0), N, Hk, st),
                          "gnm_bn_bwd_apply")
                    dZ = G
                    ws = torch.empty(int(lib.gnm_wgrad_workspace_floats(N, Hk, K)), **f32)
                    with _timed("wgrad_K%d_H%d" % (K, Hk), N=N, K=K, H=Hk):
                        check(lib.gnm_linear_wgrad(dZ.data_ptr(), dZ.stride(0), sv.x_in.data_ptr(),
                                                   sv.x_in.stride(0), N, Hk, K, ptr(sv.pro[0]) if sv.pro else None,
                                                   ptr(sv.pro[1]) if sv.pro else None, 1 if sv.pro else 0,
                                                   dW.data_ptr(), dW.stride(0), db.data_ptr(), ws.data_ptr(), st),
                              "gnm_linear_wgrad")
                    if need_dA:
                        rc2 = -2
                        if lo is not None and K == 128 and Hk == 128 and MASKED_DGRAD:
                            # dX = dZ W with the ReLU mask of the BatchNorm + ReLU below and that BatchNorm's backward
                            # sums taken in the epilogue (replaces its gnm_bn_relu_bwd_stats pass)
                            lo_part = torch.empty((int(lib.gnm_linear_grid(N)), 2, K), dtype=torch.float64, device=dev)
                            rc2 = lib.gnm_linear_dgrad_masked(
                                dZ.data_ptr(), dZ.stride(0), W.data_ptr(), W.stride(0), dA.data_ptr(), dA.stride(0), N, K,
                                Hk, lo.z.data_ptr(), lo.z.stride(0), lo.scale.data_ptr(), lo.shift.data_ptr(),
                                lo.mean.data_ptr(), lo.rstd.data_ptr(), lo_part.data_ptr(), st)
                            if rc2 == 0:
                                pre_stats = (dA, lo_part, lo_part.shape[0])
                            elif rc2 != -2:
                                check(rc2, "gnm_linear_dgrad_masked")
                        if rc2 == -2:
                            _linear_wide(dZ, W, 1, None, dA, N, Hk, K, None, None)      # dX = dZ W
                else:
                    check(rc, "gnm_linear_bwd_fused")
                if sink is None:
                    grads[wname + ".weight"], grads[wname + ".bias"] = dW, db
                incoming = dA
            # aggregation backward: d h_{l-1} = A^T (dpooled [/deg]) + (1+eps) dpooled ; d eps[l]
            if incoming is not None:
                dpooled = incoming
                want_dh = l > 0 or need_dx
                dh = torch.empty((N, F_l), **f32) if want_dh else None
                part = None
                if spec.learn_eps:
                    eps_counts[l] = int(lib.gnm_agg_num_partials(F_l, batch.n_max, B))
                    part = eps_parts[l]
                eps_ptr = P["eps"].data_ptr() + 4 * l if spec.learn_eps else None
                fused = False
                if l > 0 and want_dh and not spec.n_max:
                    # also do the layer below's outer-BatchNorm backward pass 1 in the same kernel
                    lo = saved[l - 1][2][-1]
                    use_disc = dsc1 is not None
                    Ulo = U[:, (l - 1) * H:l * H] if use_disc else None
                    dplo = dph[l - 1]
                    spart = torch.empty((B, 2, F_l), dtype=torch.float64, device=dev)
                    a = batch.arena
                    rc = -2
                    if _dense(batch, F_l, spec):
                        with _timed("agg_bwd_F%d" % F_l, F=F_l, B=B, N=N, fused_stats=1, mfma=1) as tm:
                            rc = lib.gnm_aggm_bwd_stats(
                                a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(),
                                batch.t_col_off.data_ptr(), a.bits.buf.data_ptr(), batch.t_bits_off.data_ptr(),
                                a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(), batch.node_off.data_ptr(), B,
                                batch.n_max, dpooled.data_ptr(), dpooled.stride(0), dh.data_ptr(), dh.stride(0), F_l,
                                eps_ptr, int(spec.n_avg), int(not spec.learn_eps), None, 0, ptr(part),
                                lo.z.data_ptr(), lo.z.stride(0), lo.scale.data_ptr(), lo.shift.data_ptr(),
                                lo.mean.data_ptr(), lo.rstd.data_ptr(), ptr(dplo),
                                dplo.stride(0) if dplo is not None else 0, int(spec.g_avg),
                                ptr(dsc1) if use_disc else None, ptr(Ulo), U.stride(0) if use_disc else 0,
                                ptr(inv_perm) if use_disc else None, ptr(s2sum) if use_disc else None,
                                spart.data_ptr(), st)
                            if rc != 0:
                                tm.cancel()
                        if rc == 0 and spec.learn_eps:
                            eps_counts[l] = int(lib.gnm_aggm_num_partials(F_l, B))
                    if rc == -2:
                        with _timed("agg_bwd_F%d" % F_l, F=F_l, B=B, N=N, fused_stats=1) as tm:
                            rc = lib.gnm_agg_bwd_stats(
                                a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), batch.t_rp_off.data_ptr(),
                                batch.t_col_off.data_ptr(), a.rowptr.buf.data_ptr(), batch.rp_off.data_ptr(),
                                batch.node_off.data_ptr(), B, batch.n_max, batch.nnz_max, dpooled.data_ptr(),
                                dpooled.stride(0), dh.data_ptr(), dh.stride(0), F_l, eps_ptr, int(spec.n_avg),
                                int(not spec.learn_eps), None, 0,      # h_in is recomputed from lo.z in the epilogue (d eps)
                                ptr(part), lo.z.data_ptr(), lo.z.stride(0),
                                lo.scale.data_ptr(), lo.shift.data_ptr(), lo.mean.data_ptr(), lo.rstd.data_ptr(),
                                ptr(dplo), dplo.stride(0) if dplo is not None else 0, int(spec.g_avg),
                                ptr(dsc1) if use_disc else None, ptr(Ulo), U.stride(0) if use_disc else 0,
                                ptr(inv_perm) if use_disc else None, ptr(s2sum) if use_disc else None, spart.data_ptr(),
                                st)
                            if rc != 0:
                                tm.cancel()
                    if rc == 0:
                        fused = True
                        pre_outer = (dh, spart, B)
                    elif rc != -2:
                        check(rc, "gnm_agg_bwd_stats")
                if not fused and dh is None:
                    # nothing below consumes d h: only d eps[l] = sum dpooled . h is needed -- a flat dot product
                    eps_counts[l] = int(lib.gnm_rowdot_num_partials())
                    with _timed("deps_dot_F%d" % F_l, N=N, F=F_l):
                        hin_t = hidden_tensor(h_in)
                        check(lib.gnm_rowdot_partials(dpooled.data_ptr(), dpooled.stride(0), hin_t.data_ptr(),
                                                      hin_t.stride(0), N, F_l, part.data_ptr(), st),
                              "gnm_rowdot_partials")
                elif spec.n_max:
                    cnt = _max_bwd(batch, dpooled, dh, F_l, eps_ptr, aux,
                                   hidden_tensor(h_in) if spec.learn_eps else None, part)
                    if spec.learn_eps:
                        eps_counts[l] = cnt
                elif not fused:
                    cnt = _agg(batch, dpooled, dh, F_l, eps_ptr, spec, backward=True,
                               hfwd=hidden_tensor(h_in) if spec.learn_eps else None, deps_partial=part)
                    if spec.learn_eps:
                        eps_counts[l] = cnt
                if l > 0:
                    dH_next = dh
                else:
                    dX = dh
        for j0 in range(0, len(wjobs), 32):
            jb = wjobs[j0:j0 + 32]
            nj = len(jb)
            check(lib.gnm_reduce_partials_multi(
                (C.c_void_p * nj)(*[j[0].data_ptr() for j in jb]), (C.c_void_p * nj)(*[j[1].data_ptr() for j in jb]),
                (C.c_int * nj)(*[j[1].stride(0) for j in jb]), (C.c_void_p * nj)(*[j[2].data_ptr() for j in jb]),
                (C.c_int * nj)(*[j[3] for j in jb]), (C.c_int * nj)(*[j[4] for j in jb]), nj, N, st),
                "gnm_reduce_partials_multi")
        if spec.learn_eps:
            # layers whose aggregation backward did not run (no incoming gradient) have count 0 -> d eps = 0
            check(lib.gnm_sum_partials_multi(eps_parts.data_ptr(), eps_parts.stride(0), (C.c_int * L)(*eps_counts), L,
                                             deps.data_ptr(), st), "gnm_sum_partials_multi")
            if sink is None:
                grads["eps"] = deps
        out = [None] * 9 + [dX if need_dx else None]
        for i, name in enumerate(ctx.names):
            out.append(grads.get(name) if ctx.needs_input_grad[10 + i] else None)
        return tuple(out)
