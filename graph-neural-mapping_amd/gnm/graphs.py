"""hipGraph capture of a whole training step.

A GIN training step is ~130 kernel launches plus ~60 tiny torch ops; from Python that is
~2.9 ms of host time per step, as much as the GPU needs for 512 graphs.  CapturedTrainStep
records ONE step (zero_grad -> forward -> loss -> backward) into a hipGraph through
torch.cuda.CUDAGraph -- libgnm_hip.so launches on torch's current stream, never
synchronises or allocates, so it is capturable as is -- and replays it with new inputs
copied into static buffers (batch descriptors, labels, the Infomax permutation).  The
gradient all-reduce stays outside the graph.

Everything shape-like is frozen at capture: B, N (equal-size graphs), model options.
"""
import numpy as np
import contextlib
import gc

import torch

from .arena import BatchClassMismatch, PackedStaticBatch, StaticBatch
from .core import perm_to_device



@contextlib.contextmanager
def _capture(graph, pool=None):
    """torch.cuda.graph(...) with Python's cyclic garbage collector held off for the duration of the capture.
    torch collects once on entry, but a collection can also START inside the captured region (allocation counts cross a
    threshold -- in the autograd thread as well) and free whatever cyclic garbage the warm-up passes left: a pinned
    staging tensor or an event released there is a synchronising HIP call on a capturing stream and aborts the process
    (seen once in test_run_to_run_determinism: "Fatal Python error: Aborted ... Garbage-collecting ... _backward")."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        kw = {"capture_error_mode": "thread_local"}
        if pool is not None:
            kw["pool"] = pool
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was:
            gc.enable()


class CapturedTrainStep:
    def __init__(self, model, template_batch, loss_fn, zero_grad=None, warmup=3, post_backward=None, preserve=(),
                 agg0_cache=True, gids_host=None):
        """loss_fn(c_logit, d_logit, labels) -> scalar loss.  zero_grad(): clears the
        gradient buffers (default: model.zero_grad(set_to_none=False)).  post_backward(): extra
        capturable work recorded after backward (e.g. a fused optimizer step).  The warm-up passes
        are real steps on the template batch: parameters, BatchNorm buffers and every tensor in
        `preserve` (optimizer state) are put back afterwards, so construction has no side effect.

        gids_host (round 4): the template batch's arena ids on the host, for equal-size graphs.  The step is then
        captured on a PackedStaticBatch whose ONE buffer also carries the labels and the Infomax permutation, and
        run_gids(ids, labels, perm) feeds a replay with O(B) host index arithmetic and ONE pinned upload -- instead of
        a Batch assembled by six device gathers, two uploads and a concatenation (~75 us of launches per step)."""
        self.model = model
        self.eager_fallbacks = 0           # replays that ran eagerly instead (a batch of another class: run())
        self._post = post_backward
        self._agg0_cache = agg0_cache      # False: aggregate the input features inside the step (bench.py)
        keep = [t for t in model.state_dict().values()] + list(preserve)
        snapshot = [t.clone() for t in keep]
        dev = template_batch.node_off.device
        B = template_batch.B
        self.packed = None
        if gids_host is not None and template_batch.equal_n and B > 0:
            gh = np.asarray(gids_host, dtype=np.int64)
            arena = template_batch.arena
            tb = arena._tables()
            nnz = int(tb["nnz_host"][gh].max())
            nnz_cap = max(4096, 1 << (nnz - 1).bit_length()) if nnz > 0 else 4096
            self._perm_words = (B + 1) // 2
            self.packed = PackedStaticBatch(arena, B, template_batch.n_max, template_batch.symmetric, nnz_cap,
                                            dense=template_batch.dense, iso=template_batch.iso,
                                            has_bits=template_batch.has_bits, extra_words=B + self._perm_words,
                                            two_stage=B >= 64)
            self.packed.load_gids(gh, self._pack_extra(np.zeros(B, dtype=np.int64), np.arange(B)))
            self.static = self.packed
            self.labels = self.packed.extra[:B]
            self.perm = self.packed.extra[B:].view(torch.int32)[:B]
        else:
            self.static = StaticBatch(template_batch, extra_int64=B)      # the labels ride in the same buffer
            self.labels = self.static.extra
            self.perm = torch.arange(B, dtype=torch.int32, device=dev)
        self.loss = None
        self._zero = zero_grad or (lambda: model.zero_grad(set_to_none=False))
        self._loss_fn = loss_fn
        # warm-up on a side stream (allocator, hipFuncSetAttribute, lazy inits), as torch's
        # CUDA-graph recipe prescribes, then capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (e.g. RCCL's watchdog) keep issuing HIP calls
        with _capture(self.graph):
            self.loss = self._step()
        with torch.no_grad():
            for t, s0 in zip(keep, snapshot):
                t.copy_(s0)
        torch.cuda.synchronize()
        self._arena_ptrs = self._arena_buffers()

    def _arena_buffers(self):
        """addresses the captured kernels read the graph pool from: adding graphs may re-allocate them"""
        a = self.static.batch.arena
        return (a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), a.feat.buf.data_ptr(), a.bits.buf.data_ptr(),
                tuple(sorted((k, v["buf"].data_ptr()) for k, v in a._agg0.items() if v["buf"] is not None)))

    def _step(self):
        self._zero()
        bt = self.static.batch
        X = None if self._agg0_cache else bt.arena.features(bt)
        c_logit, d_logit = self.model.forward_batch(bt, X=X, perm=self.perm)
        loss = self._loss_fn(c_logit, d_logit, self.labels)
        # (the root gradient is a cached one: loss.backward() alone fills a fresh tensor with 1.0 -- a launch per step)
        one = getattr(self, "_one", None)
        if one is None or one.shape != loss.shape or one.dtype != loss.dtype or one.device != loss.device:
            one = self._one = torch.ones_like(loss)
        loss.backward(one)
        if self._post is not None:
            self._post()
        return loss

    def _pack_extra(self, labels_host, perm_host):
        B = labels_host.shape[0]
        ex = np.zeros(B + self._perm_words, dtype=np.int64)
        ex[:B] = labels_host
        ex[B:].view(np.int32)[:B] = perm_host
        return ex

    def _check_arena(self):
        if self._agg0_cache:        # graphs added since the capture: their layer-0 aggregate must exist before a replay
            sp = self.model._spec
            self.static.batch.arena.refresh_agg0(sp.n_avg, not sp.learn_eps)
        if self._arena_buffers() != self._arena_ptrs:
            raise RuntimeError("the graph arena was re-allocated after this step was captured (graphs were added): "
                               "add every graph before building CapturedTrainStep / FusedTrainStep, or build a new one")

    def run_gids(self, gids_host, labels_host, perm=None):
        """Replay the step on the graphs with arena ids `gids_host` (host int64), labels `labels_host` (host ints):
        what a data loader hands over.  Needs the packed form (gids_host given at construction).  A selection of
        another class than the captured one runs the same step eagerly, like run()."""
        if self.packed is None:
            raise RuntimeError("run_gids needs a step captured with gids_host=...")
        from .core import check_permutation
        gh = np.asarray(gids_host, dtype=np.int64)
        B = self.packed.B
        perm = check_permutation(np.random.permutation(B) if perm is None else perm, B)
        self._check_arena()
        if not self.packed.fits(gh):
            self.eager_fallbacks += 1
            arena = self.packed.arena
            bt = arena.batch_from_gids(gh)
            lab = torch.as_tensor(np.asarray(labels_host, dtype=np.int64)).to(arena.device)
            return self._eager(bt, lab, perm)
        self.packed.load_gids(gh, self._pack_extra(np.asarray(labels_host, dtype=np.int64), perm))
        self.graph.replay()
        return self.loss

    def run(self, batch, labels, perm=None):
        """Replay the step on `batch` (same shape as the template).  perm: the
        np.random.permutation(B) of graphcnn.py:199 (drawn here if None)."""
        if perm is None:
            perm = np.random.permutation(batch.B)
        if self.packed is not None:
            # (captured on the packed buffer: a ready-made Batch is copied in vector by vector -- the general route)
            self._check_arena()
            b = self.packed.batch
            if (batch.B, batch.N, batch.n_max, batch.n_min, batch.symmetric, batch.dense, batch.iso) != \
                    (b.B, b.N, b.n_max, b.n_min, b.symmetric, b.dense, b.iso) or batch.nnz_max > b.nnz_max:
                self.eager_fallbacks += 1
                return self._eager(batch, labels, perm)
            for name in ("rp_off", "col_off", "gids", "feat_base", "bits_off") + \
                    (() if b.symmetric else ("t_rp_off", "t_col_off", "t_bits_off")):
                getattr(b, name).copy_(getattr(batch, name), non_blocking=True)
            self.labels.copy_(labels.to(torch.int64), non_blocking=True)
            perm_to_device(perm, self.perm.shape[0], self.perm.device, out=self.perm)
            self.graph.replay()
            return self.loss
        if self._agg0_cache:        # graphs added since the capture: their layer-0 aggregate must exist before a replay
            sp = self.model._spec
            batch.arena.refresh_agg0(sp.n_avg, not sp.learn_eps)
        if self._arena_buffers() != self._arena_ptrs:
            raise RuntimeError("the graph arena was re-allocated after this step was captured (graphs were added): "
                               "add every graph before building CapturedTrainStep / FusedTrainStep, or build a new one")
        try:
            self.static.load(batch, extra=labels.to(torch.int64))
        except BatchClassMismatch:
            # a batch of another class than the captured one (a pool that straddles the density threshold of the
            # matrix-core aggregation, a graph with an isolated node, another size): the same step, launched eagerly.
            # Counted (bench.py prints it) and announced once: a caller whose every batch lands here is not replaying.
            # Any other ValueError of load() is API misuse and propagates.
            self.eager_fallbacks += 1
            if self.eager_fallbacks == 1:
                import warnings
                warnings.warn("CapturedTrainStep: a batch of another class than the captured one runs eagerly "
                              "(counted in .eager_fallbacks)", RuntimeWarning, stacklevel=2)
            return self._eager(batch, labels, perm)
        perm_to_device(perm, self.perm.shape[0], self.perm.device, out=self.perm)
        self.graph.replay()
        return self.loss

    def _eager(self, batch, labels, perm):
        self._zero()
        X = None if self._agg0_cache else batch.arena.features(batch)
        c_logit, d_logit = self.model.forward_batch(batch, X=X, perm=perm)
        loss = self._loss_fn(c_logit, d_logit, labels)
        loss.backward()
        if self._post is not None:
            self._post()
        return loss


class CapturedEval:
    """GIN_InfoMaxReg.forward of B equal-size graphs in eval mode as ONE replayed hipGraph.

    The reference evaluates one graph per forward over the whole training set every epoch (main.py:49-57, 154;
    also get_latent_space, main.py:71-82).  Eager, such a forward is ~110 launches = ~1.4 ms of Python for ~0.15 ms
    of GPU work; replayed it is one pinned descriptor copy + one graph launch.  The kernels, their launch
    parameters and their order are the eager path's, so results are bitwise identical (tests/test_gpu_eval_replay.py).
    Outputs live in static buffers that the next run() overwrites: callers clone what they keep."""

    def __init__(self, model, gids_host, warmup=2):
        arena = model.arena()
        tb = arena._tables()
        gh = np.asarray(gids_host, dtype=np.int64)
        n = int(tb["n_host"][gh[0]])
        if not (tb["n_host"][gh] == n).all():
            raise ValueError("CapturedEval needs equal-size graphs (as the discriminator does, discriminator.py:24)")
        nnz = int(tb["nnz_host"][gh].max())
        nnz_cap = max(4096, 1 << (nnz - 1).bit_length()) if nnz > 0 else 4096     # launch parameters sized for this
        self.model = model
        self._fused = getattr(model, "eval_fused", False)
        self.static = PackedStaticBatch(arena, gh.shape[0], n, bool(tb["sym_host"][gh].all()), nnz_cap,
                                        dense=arena.dense_ok(gh), iso=bool(tb["iso_host"][gh].any()),
                                        has_bits=bool(tb["bits_ok_host"][gh].all()))
        self.static.load_gids(gh)
        dev = arena.device
        B = gh.shape[0]
        self.perm = torch.arange(B, dtype=torch.int32, device=dev)      # (a valid permutation from the start)
        self._params = [p for p in model.parameters()] + [b for b in model.buffers()]
        self._param_ptrs = tuple(t.data_ptr() for t in self._params)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._forward()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with _capture(self.graph):
            self.c_logit, self.d_logit, self.g_f = self._forward()
        torch.cuda.synchronize(dev)
        self._arena_ptrs = self._arena_buffers()

    def _arena_buffers(self):
        a = self.static.arena
        return (a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), a.feat.buf.data_ptr(), a.bits.buf.data_ptr(),
                tuple(sorted((k, v["buf"].data_ptr()) for k, v in a._agg0.items() if v["buf"] is not None)))

    def _forward(self):
        m = self.model
        bt = self.static.batch
        with torch.no_grad():
            if getattr(m, "eval_fused", False):
                # the evaluation encoders aggregate layer 0 themselves (a shape they decline takes the differentiable
                # path, which then does so too): no gather from the arena's layer-0 cache
                X, P0 = bt.arena.features(bt), None
            else:
                X, P0 = bt.arena.features_and_agg0(bt, m._spec.n_avg, not m._spec.learn_eps)
            return m._run(bt, X, self.perm, want_disc=True, P0=P0)

    def valid_for(self, gh):
        """still replayable for these graphs?  (shape fits, and neither the arena nor the parameters moved, nor the
        choice between the layer-by-layer kernels and the one-launch encoder)"""
        return (self.static.fits(gh) and self._arena_buffers() == self._arena_ptrs
                and tuple(t.data_ptr() for t in self._params) == self._param_ptrs
                and getattr(self.model, "eval_fused", False) == self._fused)

    def run(self, gh, perm):
        """replay on the graphs with arena ids gh (host int64 [B]) and the permutation of graphcnn.py:199;
        returns the static (c_logit, d_logit, g_f)"""
        self.static.load_gids(gh)
        if self.static.B > 1:      # (B = 1: the permutation of one graph is [0], as captured)
            # a fresh pinned tensor per call: torch's pinned-memory allocator keeps it alive until the copy has run
            perm_to_device(perm, self.perm.shape[0], self.perm.device, out=self.perm)
        self.graph.replay()
        return self.c_logit, self.d_logit, self.g_f


class CapturedTrain:
    """GIN_InfoMaxReg.forward of B equal-size graphs in TRAIN mode, and its backward, as two replayed hipGraphs -- for a
    caller that is the reference's own loop (main.py:19-47: model(batch_graph); torch losses; optimizer.zero_grad();
    loss.backward(); optimizer.step(), batch 32 by default), which this package cannot change.  Eagerly such a step is
    ~200 launches from Python (~2 ms of host time for ~0.5 ms of GPU work at B = 32); replayed it is two graph launches
    plus the caller's own loss and optimizer.

    Forward graph: the kernels of GinInfoMaxFn.forward on static batch descriptors (incl. BatchNorm running-statistics
    updates and the dropout masks, whose Philox state torch advances per replay).  Backward graph:
    torch.autograd.grad of the captured outputs against two static gradient buffers.  The gradients come back as views
    of static buffers; models/graphcnn.py hands them to autograd, which -- after the usual zero_grad(set_to_none) --
    adopts them as .grad without a copy.

    Everything shape-like is frozen: B, n, symmetry / density class, the parameters' addresses.  One forward may be
    outstanding per capture (its activations live in the graph's pool); models/graphcnn.py falls back to the eager
    path for anything else."""

    def __init__(self, model, gids_host, warmup=2):
        arena = model.arena()
        tb = arena._tables()
        gh = np.asarray(gids_host, dtype=np.int64)
        n = int(tb["n_host"][gh[0]])
        if not (tb["n_host"][gh] == n).all():
            raise ValueError("CapturedTrain needs equal-size graphs (as the discriminator does, discriminator.py:24)")
        nnz = int(tb["nnz_host"][gh].max())
        nnz_cap = max(4096, 1 << (nnz - 1).bit_length()) if nnz > 0 else 4096
        self.model = model
        self.static = PackedStaticBatch(arena, gh.shape[0], n, bool(tb["sym_host"][gh].all()), nnz_cap,
                                        dense=arena.dense_ok(gh), iso=bool(tb["iso_host"][gh].any()),
                                        has_bits=bool(tb["bits_ok_host"][gh].all()))
        self.static.load_gids(gh)
        dev = arena.device
        B = gh.shape[0]
        # a PERMUTATION from the start: the warm-up passes run the backward, whose shuffled-branch term indexes by the
        # inverse permutation (zeros here left it mostly unwritten -> wild reads: a GPU memory fault in the first version)
        self.perm = torch.arange(B, dtype=torch.int32, device=dev)
        self._params = [p for p in model.parameters()]
        self._req = [p for p in self._params if p.requires_grad]
        self._tracked = self._params + [b for b in model.buffers()]
        self._ptrs = tuple(t.data_ptr() for t in self._tracked)
        self._req_mask = tuple(p.requires_grad for p in self._params)
        self.outstanding = None        # weak reference to the hold of a replayed forward that has not been backpropagated
        self.gen = 0                   # number of forward replays so far (claim() / _TrainReplayFn.backward)
        keep = [b for b in model.buffers()]                # BatchNorm statistics: the warm-up passes must not count
        snapshot = [t.clone() for t in keep]
        # The gradients land in ONE persistent buffer of this object (ordinary allocation, outside the graphs' memory
        # pool, so no replay's temporaries can sit on it): the backward kernels write them there directly
        # (GinSpec.grad_sink) and autograd is handed views of it.
        names = [nm for nm, _ in model.named_parameters()]
        self._flat = torch.zeros(sum(p.numel() for p in self._params), dtype=torch.float32, device=dev)
        own_sink, off = {}, 0
        for nm, p in zip(names, self._params):
            own_sink[nm] = self._flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        sink = model._spec.grad_sink
        try:
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                # which parameters get a gradient at all (eps does not when learn_eps is False): one plain pass
                model._spec.grad_sink = None
                c, d, _ = self._forward()
                plain = torch.autograd.grad((c, d), self._req, (torch.zeros_like(c), torch.zeros_like(d)), allow_unused=True)
                used = iter([g is not None for g in plain])
                self._has_grad = [bool(next(used)) if p.requires_grad else False for p in self._params]
                del plain
                model._spec.grad_sink = own_sink
                for _ in range(warmup):
                    c, d, _ = self._forward()
                    torch.autograd.grad((c, d), self._req, (torch.zeros_like(c), torch.zeros_like(d)), allow_unused=True)
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            self.fwd_graph = torch.cuda.CUDAGraph()
            with _capture(self.fwd_graph):
                self.c_logit, self.d_logit, _ = self._forward()
            self.dC = torch.zeros_like(self.c_logit)
            self.dD = torch.zeros_like(self.d_logit)
            self.bwd_graph = torch.cuda.CUDAGraph()
            with _capture(self.bwd_graph, pool=self.fwd_graph.pool()):
                torch.autograd.grad((self.c_logit, self.d_logit), self._req, (self.dC, self.dD), allow_unused=True)
        finally:
            model._spec.grad_sink = sink
            # the warm-up (and captured) passes ran real forwards: BatchNorm statistics and counters go back to what
            # they were -- also when the capture failed and the caller falls back to the eager path
            with torch.no_grad():
                for t, s0 in zip(keep, snapshot):
                    t.copy_(s0)
        self.grads = [own_sink[nm] if has else None for nm, has in zip(names, self._has_grad)]
        torch.cuda.synchronize(dev)
        self._arena_ptrs = self._arena_buffers()
        # The captured kernels read the graph pool at the addresses of capture time.  A forward re-checks them
        # (valid_for), but the arena may GROW between a replayed forward and its backward (another batch's graphs added
        # in between: the arena then moves to bigger buffers and drops the old ones) -- so this capture keeps the
        # buffers it was recorded on alive: they stay complete for the graphs it can be replayed on.
        a = arena
        self._arena_refs = (a.rowptr.buf, a.col.buf, a.feat.buf, a.bits.buf,
                            tuple(v["buf"] for v in a._agg0.values() if v["buf"] is not None))
        self._grad_ptrs = frozenset(g.data_ptr() for g in self.grads if g is not None)
        self._flat.zero_()

    _arena_buffers = CapturedEval._arena_buffers

    def _forward(self):
        m = self.model
        bt = self.static.batch
        X, P0 = bt.arena.features_and_agg0(bt, m._spec.n_avg, not m._spec.learn_eps)
        # (the same score kernel variant as an eager train-mode forward -- the one that also leaves the discriminator's
        #  backward reductions, unused under torch's own losses -- so that replayed and eager outputs are the same bits)
        return m._run(bt, X, self.perm, want_disc=True, P0=P0)

    def valid_for(self, gh):
        m = self.model
        return (self.static.fits(gh) and self._arena_buffers() == self._arena_ptrs
                and tuple(t.data_ptr() for t in self._tracked) == self._ptrs
                and tuple(p.requires_grad for p in self._params) == self._req_mask
                and m._spec.grad_sink is None and m._spec.sync_bn is None)

    class _Hold:
        """lives on the autograd node of one replayed forward (weakly referenced from here)"""
        __slots__ = ("__weakref__",)

    def claim(self):
        """called by the autograd Function right after forward(): (generation of that replay, an object whose lifetime
        is the autograd node's -- while it is alive and backward has not run, the activations are taken: busy())"""
        import weakref
        h = CapturedTrain._Hold()
        self.outstanding = weakref.ref(h)
        return self.gen, h

    def busy(self):
        """a replayed forward whose autograd node is still alive (either output, or anything computed from them) and
        not yet backpropagated owns the captured activations"""
        o = self.outstanding
        return o is not None and o() is not None

    def _protect_grads(self):
        """A .grad that still aliases the gradient buffer (adopted by autograd after the last backward and not cleared
        since: gradient accumulation) must own its values before the backward replay rewrites the buffer.  (The forward
        replay cannot touch it: the buffer lives outside the graphs' memory pool -- in a first version the gradients
        were outputs of the backward graph inside the shared pool, where the next forward replay's temporaries
        overwrote them.)"""
        for p in self._req:
            g = p.grad
            if g is not None and g.data_ptr() in self._grad_ptrs:
                p.grad = g.clone()

    def forward(self, gh, perm):
        self.gen += 1
        self.static.load_gids(gh)
        if self.static.B > 1:
            perm_to_device(perm, self.perm.shape[0], self.perm.device, out=self.perm)
        self.fwd_graph.replay()
        return self.c_logit, self.d_logit

    def backward(self, dC, dD):
        """replays the backward on the outstanding forward; returns one gradient per parameter (None where unused) as
        FRESH views of the static buffers (a tensor nobody else references is what autograd may adopt as .grad)"""
        self._protect_grads()
        if dC is None:
            self.dC.zero_()
        else:
            self.dC.copy_(dC)
        if dD is None:
            self.dD.zero_()
        else:
            self.dD.copy_(dD)
        self.bwd_graph.replay()
        self.outstanding = None
        return [g.view_as(g) if g is not None else None for g in self.grads]
