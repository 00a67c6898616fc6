"""GIN_InfoMaxReg on MI355X: the reference's model interface over the HIP hot path.

Drop-in for /root/reference models/graphcnn.py (class at :12, constructor :13, forward
:194, compute_saliency :254): `from models.graphcnn import *` in the reference's main.py
(main.py:9) resolves to this file when main.py is run from graph-neural-mapping_amd/.
Constructor arguments, submodule names / creation order (=> identical seeded init and
state_dict keys), numpy-RNG consumption (one np.random.permutation(B) per forward,
graphcnn.py:199), return values and error behaviour follow the reference.

What differs is how the work is done: graphs are converted once to a device-resident CSR
arena (gnm/arena.py), and everything from `X_concat` on runs in libgnm_hip.so through one
autograd.Function with a hand-written backward (gnm/core.py).  There is no CPU or eager
fallback -- on a CPU device forward() raises.  neighbor_pooling_type == "max" (outside
BASELINE.json's north_star) runs through the same engine with
csrc/maxpool.hip in place of the aggregation kernels (gnm/maxnb.py builds its neighbour
lists from graph.neighbors, as graphcnn.py:55-81 does); only the hipGraph replays and the
layer-0 cache are not offered for it.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
for _p in (_PKG, _HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from mlp import MLP  # noqa: E402
from discriminator import Discriminator  # noqa: E402
from gnm.arena import GraphArena  # noqa: E402
from gnm.core import (DISC_UNIT, DiscUnit, GinInfoMaxFn, GinSpec, eval_forward_fused,  # noqa: E402
                      eval_fused_ok)

__all__ = ["GIN_InfoMaxReg", "GraphCNN", "MLP", "Discriminator"]


class _LazyEvalGrad(torch.autograd.Function):
    """Outputs of a replayed eval forward, attached to the parameters: backward() recomputes the forward eagerly
    (same graphs, same permutation) and differentiates that.  Nobody in the reference backpropagates through an
    eval-mode forward; this only keeps doing so correct."""

    @staticmethod
    def forward(ctx, model, gh, perm, c_logit, d_logit, *params):
        ctx.model, ctx.gh, ctx.perm = model, gh, perm
        return c_logit.view_as(c_logit), d_logit.view_as(d_logit)

    @staticmethod
    def backward(ctx, dC, dD):
        m = ctx.model
        names, tensors, _ = m._param_lists()
        with torch.enable_grad():
            batch = m.arena().batch_from_gids(ctx.gh)
            X, P0 = batch.arena.features_and_agg0(batch, m._spec.n_avg, not m._spec.learn_eps)
            sink, m._spec.grad_sink = m._spec.grad_sink, None          # plain autograd gradients here
            try:
                c, d, _ = m._run(batch, X, ctx.perm, want_disc=True, P0=P0, allow_fused=False)
                outs, gouts = [], []
                for o, g in ((c, dC), (d, dD)):
                    if g is not None:
                        outs.append(o)
                        gouts.append(g)
                req = [t for t in tensors if t.requires_grad]
                grads = torch.autograd.grad(outs, req, gouts, allow_unused=True)
            finally:
                m._spec.grad_sink = sink
        it = iter(grads)
        return (None, None, None, None, None) + tuple(next(it) if t.requires_grad else None for t in tensors)


class _LazyEvalGradBatch(torch.autograd.Function):
    """Outputs of an eval-mode forward that ran on the evaluation encoder (no autograd graph), attached to the
    parameters like _LazyEvalGrad: backward() runs the differentiable forward on the same batch and differentiates that."""

    @staticmethod
    def forward(ctx, model, batch, X, P0, perm, c_logit, d_logit, *params):
        ctx.model, ctx.batch, ctx.X, ctx.P0, ctx.perm = model, batch, X, P0, perm
        return c_logit.view_as(c_logit), d_logit.view_as(d_logit)

    @staticmethod
    def backward(ctx, dC, dD):
        m = ctx.model
        names, tensors, _ = m._param_lists()
        with torch.enable_grad():
            sink, m._spec.grad_sink = m._spec.grad_sink, None          # plain autograd gradients here
            try:
                c, d, _ = m._run(ctx.batch, ctx.X, ctx.perm, want_disc=True, P0=ctx.P0, allow_fused=False)
                outs, gouts = [], []
                for o, g in ((c, dC), (d, dD)):
                    if g is not None and o.requires_grad:
                        outs.append(o)
                        gouts.append(g)
                req = [t for t in tensors if t.requires_grad]
                grads = torch.autograd.grad(outs, req, gouts, allow_unused=True) if outs else [None] * len(req)
            finally:
                m._spec.grad_sink = sink
        it = iter(grads)
        return (None,) * 7 + tuple(next(it) if t.requires_grad else None for t in tensors)


class _TrainReplayFn(torch.autograd.Function):
    """A train-mode forward replayed from a captured hipGraph (gnm/graphs.py CapturedTrain), attached to the parameters
    so that the caller's loss.backward() replays the captured backward.  Outputs are copies (the static buffers belong
    to the next replay)."""

    @staticmethod
    def forward(ctx, cap, gh, perm, *params):
        c_logit, d_logit = cap.forward(gh, perm)
        ctx.cap = cap
        # The capture's activations belong to THIS forward until its backward has run.  `hold` lives exactly as long as
        # this autograd node -- i.e. while either output or anything computed from them is alive (a weak reference to
        # c_logit alone, as in round 3, let `_, d = model(b)` look finished) -- and the generation number lets backward
        # prove that no later replay has overwritten them.
        ctx.gen, ctx.hold = cap.claim()
        ctx.set_materialize_grads(False)
        return c_logit.clone(), d_logit.clone()

    @staticmethod
    def backward(ctx, dC, dD):
        if ctx.gen != ctx.cap.gen:
            raise RuntimeError("train replay: the captured activations of this forward were overwritten by a later "
                               "forward before its backward ran (generation %d, now %d); set model.train_replay = False "
                               "for this pattern" % (ctx.gen, ctx.cap.gen))
        return (None, None, None) + tuple(ctx.cap.backward(dC, dD))


class GIN_InfoMaxReg(nn.Module):
    def __init__(self, num_layers, num_mlp_layers, input_dim, hidden_dim, output_dim, final_dropout, learn_eps,
                 graph_pooling_type, neighbor_pooling_type, device):
        super().__init__()
        self._check_kernel_limits(num_layers, input_dim, hidden_dim)
        # creation order follows graphcnn.py:29-52 so torch.manual_seed(s) yields the same weights
        self.disc = Discriminator(hidden_dim * num_layers)
        self.sigm = nn.Sigmoid()
        self.relu = nn.ReLU()
        self.final_dropout = final_dropout
        self.device = device
        self.num_layers = num_layers
        self.num_mlp_layers = num_mlp_layers
        self.graph_pooling_type = graph_pooling_type
        self.neighbor_pooling_type = neighbor_pooling_type
        self.learn_eps = learn_eps
        self.eps = nn.Parameter(torch.zeros(num_layers))
        self.mlps = nn.ModuleList()
        self.batch_norms = nn.ModuleList()
        self.linears_prediction = nn.ModuleList()
        for layer in range(num_layers):
            self.mlps.append(MLP(num_mlp_layers, input_dim if layer == 0 else hidden_dim, hidden_dim, hidden_dim))
            self.batch_norms.append(nn.BatchNorm1d(hidden_dim))
            self.linears_prediction.append(nn.Linear(hidden_dim, output_dim))
        self._spec = GinSpec(num_layers, num_mlp_layers, learn_eps, graph_pooling_type, neighbor_pooling_type)
        self._arena = None
        self._plist = None
        # eval-mode forwards of small batches are replayed from captured hipGraphs (gnm/graphs.py CapturedEval):
        # {(B, n): CapturedEval}, a few entries; eval_replay = False turns it off
        self.eval_replay = True
        self._eval_cache = {}
        # train-mode forwards (and their backwards) of small batches are replayed too (gnm/graphs.py CapturedTrain):
        # what makes the reference's own loop (main.py:19-47, batch 32) GPU-bound instead of launch-bound.
        # train_replay = False turns it off
        self.train_replay = True
        self._train_cache = {}
        # eval_fused = True: eval-mode forwards under torch.no_grad() (incl. the replayed ones) run the L layers as ONE
        # encoder launch, a workgroup per graph (csrc/evalfwd.hip), when the shape allows.  Off by default: measured
        # at one 400-node graph per forward it is SLOWER than the replayed layer-by-layer kernels (0.245 vs 0.188 ms per
        # graph, gpurun_out/r03i_time_eval.log) -- one CU runs a graph's whole chain (MFMA floor ~57 us + 50 barriers)
        # where the ~110 replayed launches (~1.4 us apiece) each spread over several CUs.  DESIGN.md section 6.
        # eval_fused = "layers": one launch per LAYER with a workgroup per 32-row block of every graph
        # (csrc/evallayer.hip): 13 CUs work on a 400-node graph, ~10 launches per forward.  0.130 ms per graph -- the
        # default.  False: the training kernels in eval mode.
        self.eval_fused = "layers"

    @staticmethod
    def _check_kernel_limits(num_layers, input_dim, hidden_dim):
        """Shapes the HIP kernels are built for (narrower than the reference, which takes anything torch does):
        said here, at construction, rather than as GNM_ERR_BAD_ARG from the first forward."""
        from gnm._cabi import lib
        if hidden_dim < 4 or hidden_dim > 128 or hidden_dim % 4:
            raise ValueError("hidden_dim=%d: the MI355X kernels take a multiple of 4 in [4, 128] (BatchNorm / Linear "
                             "column tiles; csrc/norm.hip, csrc/linear.hip)" % hidden_dim)
        if num_layers < 1 or num_layers > 16:
            raise ValueError("num_layers=%d: 1..16 GIN layers are supported (csrc/disc.hip GNM_MAX_LAYERS)" % num_layers)
        kmax = int(lib.gnm_linear_max_k(hidden_dim))
        if input_dim < 1 or input_dim > kmax:
            raise ValueError("input_dim=%d with hidden_dim=%d: the first Linear keeps its [input_dim x hidden_dim] "
                             "weight in LDS, which bounds input_dim to %d (csrc/linear.hip)"
                             % (input_dim, hidden_dim, kmax))

    # ------------------------------------------------------------------ plumbing
    def arena(self):
        """Device-resident graph store; created lazily on the parameters' device."""
        dev = self.eps.device
        if self._arena is None or self._arena.device != dev:
            self._arena = GraphArena(dev)
        return self._arena

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() may replace buffer tensors: drop the cached lists and captured graphs
        self._plist = None
        self._eval_cache = {}
        self._train_cache = {}
        return super()._apply(fn, *args, **kwargs)

    def _param_lists(self):
        """(names, parameters, buffers) in state_dict order, cached: walking the module
        tree on every forward costs ~0.4 ms of Python."""
        pl = getattr(self, "_plist", None)
        if pl is None:
            names, tensors = zip(*self.named_parameters())
            pl = self._plist = (names, tensors, dict(self.named_buffers()))
        return pl

    def _run(self, batch, X, perm, want_disc, P0=None, hand_over=True, allow_fused=True):
        names, tensors, buffers = self._param_lists()
        if (allow_fused and not self.training and self.eval_fused and not X.requires_grad
                and not (P0 is not None and P0.requires_grad)):
            # evaluation (the replayed eval forward, callers under torch.no_grad(), and -- so that every eval-mode
            # forward of a model gives the same bits -- callers in grad mode too): the evaluation encoder
            # (csrc/evallayer.hip / csrc/evalfwd.hip) instead of ~20 training-kernel launches per layer.  It records
            # no autograd graph; in grad mode the outputs are attached lazily (_LazyEvalGradBatch), as the reference's
            # eval outputs carry one (main.py:54 detaches them).  Saliency (X.requires_grad) takes the path below.
            P = dict(zip(names, tensors))
            P.update(buffers)
            if eval_fused_ok(self._spec, batch, X, P, self.eval_fused):
                with torch.no_grad():
                    c_logit, d_logit, g_f = eval_forward_fused(self._spec, batch, perm, P, X, want_disc, self.eval_fused)
                if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
                    c_logit, d_logit = _LazyEvalGradBatch.apply(self, batch, X, P0, perm, c_logit, d_logit, *tensors)
                return c_logit, d_logit, g_f
        hold = None
        if want_disc and self.training and DISC_UNIT and hand_over and torch.is_grad_enabled():
            # let the score kernel leave the backward's reductions for the reference's BCE loss (gnm/core.py DiscUnit);
            # only a loss that recognises the hand-over on d_logit (gnm.train.infomax_loss) makes use of it
            hold = want_disc = DiscUnit()
        out = GinInfoMaxFn.apply(self._spec, batch, perm, names, buffers, self.training, float(self.final_dropout),
                                 want_disc, P0, X, *tensors)
        if hold is not None and hold.unit is not None:
            out[1]._gnm_disc_unit = hold
        return out

    def forward_batch(self, batch, X=None, perm=None, latent=False):
        """forward() for an already assembled gnm.arena.Batch (what bench.py and the
        data-parallel driver call: no per-graph Python work)."""
        if perm is None:
            perm = np.random.permutation(batch.B)                             # graphcnn.py:199
        P0 = None
        if X is None and self._spec.n_max:
            X = batch.arena.features(batch)
        elif X is None:
            # layer 0's A X [/deg] does not depend on the parameters: gathered from the arena's per-graph cache
            X, P0 = batch.arena.features_and_agg0(batch, self._spec.n_avg, not self._spec.learn_eps)
        c_logit, d_logit, g_f = self._run(batch, X, perm, want_disc=True, P0=P0)
        if latent:
            return g_f.detach().cpu().numpy()                                  # graphcnn.py:248-249
        return c_logit, d_logit

    @torch.no_grad()
    def predict(self, graphs, batch_size=256, latent=False):
        """Evaluation over many graphs in batches (not in the reference: its test() / get_latent_space() call
        forward([g]) once per graph, main.py:49-57,71-82, which is launch-bound).  Eval mode -- BatchNorm uses its
        running statistics, so batching does not change a graph's result.  Returns c_logit [len(graphs), C], or
        the [len(graphs), L*H] latent array when latent=True.  Graphs of a batch must have equal node counts
        (discriminator.py:24), as in forward()."""
        was_training = self.training
        self.eval()
        try:
            out = []
            for i in range(0, len(graphs), batch_size):
                chunk = graphs[i:i + batch_size]
                r = self.forward(chunk, latent=latent)
                out.append(r if latent else r[0])
            if latent:
                return np.concatenate(out, 0) if out else np.zeros((0, 0), dtype=np.float32)
            return torch.cat(out, 0) if out else torch.zeros((0, 0), device=self.eps.device)
        finally:
            self.train(was_training)

    # ------------------------------------------------------------------ evaluation replay
    EVAL_REPLAY_MAX_B = 64          # larger eval batches are GPU-bound anyway (and hold more captured activations)
    EVAL_REPLAY_ENTRIES = 6

    def _forward_eval_replay(self, batch_graph, latent):
        """forward() in eval mode for a small batch of equal-size graphs: the eager path's kernels replayed from
        a captured hipGraph (bitwise the same results, ~0.1 ms of host time instead of ~1.4 ms).  None when the batch
        does not qualify -- the caller then takes the eager path."""
        arena = self.arena()
        if arena.device.type != "cuda":
            return None
        gh = np.asarray(arena.add_many(batch_graph), dtype=np.int64)
        tb = arena._tables()
        n = int(tb["n_host"][gh[0]])
        B = int(gh.shape[0])
        if B > 1 and not (tb["n_host"][gh] == n).all():
            return None
        from gnm.graphs import CapturedEval
        arena.refresh_agg0(self._spec.n_avg, not self._spec.learn_eps)       # the replay only gathers from this cache
        key = (B, n)
        ce = self._eval_cache.get(key)
        rng_state = np.random.get_state() if (ce is None or not ce.valid_for(gh)) else None
        perm = np.random.permutation(B)                                       # graphcnn.py:199, consumed as always
        if rng_state is not None:
            if len(self._eval_cache) >= self.EVAL_REPLAY_ENTRIES:
                self._eval_cache.pop(next(iter(self._eval_cache)))
            try:
                with torch.cuda.device(arena.device):
                    ce = self._eval_cache[key] = CapturedEval(self, gh)
            except Exception as e:          # e.g. a user hook that synchronises inside the forward: stay eager
                import warnings
                warnings.warn("eval-mode hipGraph capture failed (%s: %s); evaluating eagerly from now on"
                              % (type(e).__name__, e))
                self.eval_replay = False
                np.random.set_state(rng_state)
                return None
        c_logit, d_logit, g_f = ce.run(gh, perm)
        if latent:
            return g_f.cpu().numpy()                                          # graphcnn.py:248-249
        c_logit, d_logit = c_logit.clone(), d_logit.clone()                  # the static buffers are reused
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # the reference's outputs carry an autograd graph even in eval mode (main.py:54 detaches them):
            # keep that contract lazily -- a backward through these outputs re-runs the eager forward
            names, tensors, _ = self._param_lists()
            return _LazyEvalGrad.apply(self, gh, perm, c_logit, d_logit, *tensors)
        return c_logit, d_logit

    # ------------------------------------------------------------------ training replay
    TRAIN_REPLAY_MAX_B = 128        # beyond this a step is GPU-bound from Python too (and activations get large)
    TRAIN_REPLAY_ENTRIES = 3

    def _forward_train_replay(self, batch_graph):
        """forward() in train mode for a small batch of equal-size graphs: forward and backward replayed from captured
        hipGraphs (same kernels, same order as the eager path).  None when the batch does not qualify -- other shapes,
        a forward still outstanding on the capture, a gradient sink / cross-rank BatchNorm installed, parameters that
        moved -- and the caller then takes the eager path."""
        arena = self.arena()
        sp = self._spec
        if arena.device.type != "cuda" or sp.grad_sink is not None or sp.sync_bn is not None or sp.keep_hidden:
            return None
        gh = np.asarray(arena.add_many(batch_graph), dtype=np.int64)
        tb = arena._tables()
        n = int(tb["n_host"][gh[0]])
        B = int(gh.shape[0])
        if not (tb["n_host"][gh] == n).all():
            return None
        from gnm.graphs import CapturedTrain
        arena.refresh_agg0(sp.n_avg, not sp.learn_eps)
        key = (B, n)
        ct = self._train_cache.get(key)
        if ct is not None and ct.busy():
            return None                                   # its activations belong to a forward not yet backpropagated
        if ct is None or not ct.valid_for(gh):
            if len(self._train_cache) >= self.TRAIN_REPLAY_ENTRIES and key not in self._train_cache:
                self._train_cache.pop(next(iter(self._train_cache)))
            rng_state = np.random.get_state()
            try:
                with torch.cuda.device(arena.device):
                    ct = self._train_cache[key] = CapturedTrain(self, gh)
            except Exception as e:          # e.g. a hook that synchronises inside the forward: stay eager
                import warnings
                warnings.warn("train-mode hipGraph capture failed (%s: %s); training eagerly from now on"
                              % (type(e).__name__, e))
                self.train_replay = False
                self._train_cache.pop(key, None)
                np.random.set_state(rng_state)
                return None
        perm = np.random.permutation(B)                                       # graphcnn.py:199, consumed as always
        names, tensors, _ = self._param_lists()
        c_logit, d_logit = _TrainReplayFn.apply(ct, gh, perm, *tensors)
        return c_logit, d_logit

    # ------------------------------------------------------------------ reference API
    def _batch_of(self, batch_graph):
        batch = self.arena().batch(batch_graph)
        if self._spec.n_max:
            from gnm.maxnb import MaxNeighbours
            batch.maxnb = MaxNeighbours(batch_graph, not self.learn_eps, self.arena().device)   # graphcnn.py:55-81
        return batch

    def forward(self, batch_graph, latent=False):
        if (not self.training and self.eval_replay and not self._spec.n_max
                and 0 < len(batch_graph) <= self.EVAL_REPLAY_MAX_B):
            out = self._forward_eval_replay(batch_graph, latent)
            if out is not None:
                return out
        if (self.training and self.train_replay and not latent and not self._spec.n_max and torch.is_grad_enabled()
                and 0 < len(batch_graph) <= self.TRAIN_REPLAY_MAX_B):
            out = self._forward_train_replay(batch_graph)
            if out is not None:
                return out
        return self.forward_batch(self._batch_of(batch_graph), latent=latent)

    def compute_saliency(self, batch_graph, cls):
        self.eval()
        self.zero_grad()
        assert len(batch_graph) == 1                                           # graphcnn.py:257
        batch = self._batch_of(batch_graph)
        X = batch.arena.features(batch).detach().requires_grad_()
        score, _, _ = self._run(batch, X, np.zeros(1, dtype=np.int64), want_disc=False)
        predicting_class = torch.zeros([1, 2], device=X.device)                # two classes hard-coded (:263)
        predicting_class[0, cls] = 1
        score.backward(predicting_class)
        return X.grad


GraphCNN = GIN_InfoMaxReg  # the name BASELINE.json's north_star uses
