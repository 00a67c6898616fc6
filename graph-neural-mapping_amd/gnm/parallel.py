"""Data-parallel driver: one process per GPU, batch sharded by graph, ONE all-reduce of a
flat fp32 gradient buffer per step (RCCL over xGMI when the backend is "nccl").

The reference is single-process (main.py:126); this is the MI355X-native addition of
SURVEY.md section 8(e).  A graph is never split: rank r owns graphs [r*B/W, (r+1)*B/W) of
every global batch, no edge crosses ranks, so the only exchange step is the gradient
sum.  The model has 142,288 parameters (569 KB) at H=64: the collective is latency-bound,
so all gradients live in one contiguous buffer and a single call moves them.
BatchNorm statistics are per rank (standard DDP semantics).
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class FlatParams:
    """Re-homes every parameter and its .grad as views of two flat fp32 buffers."""

    def __init__(self, module):
        params = [p for p in module.parameters()]
        self.params = params
        total = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p.data)
            p.grad = self.flat_grad[off:off + n].view_as(p.data)
            off += n
        self.total = total

    def attach_grads(self):
        """Make every parameter's .grad the view of the flat gradient buffer again.  optimizer.zero_grad() and
        module.zero_grad() default to set_to_none=True (and GIN_InfoMaxReg.compute_saliency calls the latter), and
        autograd may replace .grad: an optimizer that finds .grad None silently skips the parameter."""
        off = 0
        base = self.flat_grad.data_ptr()
        for p in self.params:
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                p.grad = self.flat_grad[off:off + n].view_as(p.data)
            off += n

    def zero_grad(self):
        self.flat_grad.zero_()
        self.attach_grads()


class _BnSync:
    """GinSpec.sync_bn: SUM over the data-parallel group of the BatchNorm column sums (2*H doubles per
    BatchNorm, forward and backward: 20 tiny collectives per step at L = 5, m = 2) -- SURVEY.md 8(e)."""

    def __init__(self, group):
        self.group = group

    def all_reduce(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def global_count(self, n, device):
        t = torch.tensor([n], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return int(t.item())            # one host sync per forward; sync_bn is the parity mode, not the fast one


class DataParallelGIN:
    """Wraps a GIN_InfoMaxReg replica.  broadcast_parameters() once, then per step:
    zero_grad(); loss(forward(local shard)).backward(); allreduce_gradients().

    sync_bn=True: train-mode BatchNorm uses union-batch statistics (all ranks then reproduce a single
    process on the whole batch, for equal shard sizes); default False = per-rank statistics, the usual
    DistributedDataParallel semantics and the scaling configuration."""

    def __init__(self, model, process_group=None, direct_grads=True, sync_bn=False):
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.fp = FlatParams(model)
        self._stage = None          # pinned host staging buffer of the gloo-with-device-tensors path
        self._avg_ok = None         # does this RCCL take ReduceOp.AVG?  (None: not tried yet)
        self.avg_fallback_reason = None
        # test hook: run the collective even at world size 1 (a one-GPU box can then execute the RCCL code path:
        # communicator init, the AVG all-reduce, its capture into the step's hipGraph)
        self.force_collective = dist.is_initialized() and os.environ.get("GNM_DP_FORCE_COLLECTIVE") == "1"
        # GIN_InfoMaxReg: let the backward kernels write gradients straight into the flat buffer
        # (overwrite semantics: every step produces every gradient, so no zeroing and no
        # AccumulateGrad adds are needed; see GinSpec.grad_sink)
        self.direct = False
        spec = getattr(model, "_spec", None)
        if direct_grads and spec is not None:
            sink, off = {}, 0
            names = [n for n, _ in model.named_parameters()]
            for n, p in zip(names, self.fp.params):
                sink[n] = self.fp.flat_grad[off:off + p.numel()].view_as(p.data)
                off += p.numel()
            spec.grad_sink = sink
            self.direct = True
        if sync_bn:
            if spec is None or self.world == 1:
                raise ValueError("sync_bn needs a GIN_InfoMaxReg model and an initialised process group (world > 1)")
            spec.sync_bn = _BnSync(process_group)

    def broadcast_parameters(self, src=0):
        if self.world > 1:
            dist.broadcast(self.fp.flat, src=src, group=self.group)
            for b in self.model.buffers():
                dist.broadcast(b, src=src, group=self.group)

    def shard(self, global_items):
        """This rank's contiguous slice of a global batch (list or index array)."""
        B = len(global_items)
        per = B // self.world
        if per * self.world != B:
            raise ValueError("global batch %d is not divisible by world size %d" % (B, self.world))
        return global_items[self.rank * per:(self.rank + 1) * per]

    def zero_grad(self):
        if self.direct:
            # every gradient is overwritten by the next backward, so nothing is zeroed; but the .grad views must
            # be in place: a stock optimizer's zero_grad() (set_to_none=True) detaches them and would then skip
            # every parameter, because the sink backward hands autograd None
            self.fp.attach_grads()
            return
        self.fp.zero_grad()

    def allreduce_gradients(self, async_op=False):
        """Mean over ranks of the flat gradient buffer (the mean-loss convention of main.py:34-37 applied to the
        union batch).  RCCL ("nccl"): ONE collective launch, the 1/W folded into it (ReduceOp.AVG, ncclAvg) -- round 2
        followed a SUM with a separate g.div_(W) launch.  A build of RCCL that refuses AVG falls back to that pair,
        once, and remembers.  Nothing here synchronises the host, so the call is hipGraph-capturable on "nccl"
        (bench.py captures it inside the step's graph when the capture succeeds)."""
        self.fp.attach_grads()      # callers that only ran optimizer.zero_grad(): see zero_grad()
        self.last_needs_scale = False
        if self.world == 1 and not self.force_collective:
            return None
        g = self.fp.flat_grad
        if async_op and not (g.is_cuda and dist.get_backend(self.group) == "gloo"):
            # ONE contract for the asynchronous form (AVG on one path and un-scaled sums on the other left
            # the caller guessing): always the SUM; the caller scales by 1 / world after work.wait()
            # (FusedAdam.set_grad_scale).  `last_needs_scale` says so.
            self.last_needs_scale = True
            return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if g.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo has no device path worth the name: handing it a CUDA tensor makes it stage through pageable
            # memory on internal streams (measured 8 ms for 569 KB, and 300 ms when a hipGraph replay was still in
            # flight on a GPU shared by two ranks).  Stage explicitly through ONE pinned buffer instead.  This is
            # the CPU-test / shared-GPU test-hook path; RCCL ("nccl") reduces the device buffer in place.
            if self._stage is None:
                self._stage = torch.empty(g.numel(), dtype=g.dtype).pin_memory()
            self._stage.copy_(g, non_blocking=True)
            torch.cuda.current_stream(g.device).synchronize()
            dist.all_reduce(self._stage, op=dist.ReduceOp.SUM, group=self.group)
            self._stage.div_(self.world)
            g.copy_(self._stage, non_blocking=True)
            return None
        if self._avg_ok is not False and dist.get_backend(self.group) == "nccl":
            try:
                dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
                self._avg_ok = True
                return None
            except (RuntimeError, ValueError) as e:
                if self._avg_ok:            # it worked before: this is a real failure, not a missing feature
                    raise
                self._avg_ok = False
                self.avg_fallback_reason = "%s: %s" % (type(e).__name__, e)
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        g.div_(self.world)
        return None


def choose_launch_mode(candidates, measure, agree=None, errors=None):
    """Pick how a multi-rank step is launched.  candidates: mode names in order of preference (bench.py: "graph+cc" =
    hipGraph replay with the all-reduce captured inside, "graph" = replay + collective behind it, "eager");
    measure(name) -> seconds per step with that mode, already MAX-reduced over the ranks so that every rank takes
    the same decision.  A candidate whose measurement raises is dropped (e.g. a captured collective that fails at
    replay); ties go to the earlier candidate.  Returns (mode, {name: seconds}).

    measure() contains collectives, so a failure on ONE rank must take the candidate out on ALL of them, or the others
    sit in its all-reduce while this one has moved on: agree(ok) -> bool is the MIN over the ranks of a success flag
    (bench.py: a one-int all-reduce), called once per candidate after its measurement.  The exception text of a dropped
    candidate is kept in `errors` (a dict, name -> text).  A failure that is a GPU fault rather than a declined
    capture is not survivable: it is re-raised so that the process exits non-zero."""
    times = {}
    for name in candidates:
        ok, t = True, float("nan")
        try:
            t = float(measure(name))
        except Exception as e:
            text = "%s: %s" % (type(e).__name__, e)
            if errors is not None:
                errors[name] = text
            low = text.lower()
            if any(k in low for k in ("memory access fault", "hiperrorillegaladdress", "illegal memory access",
                                      "hiperrorlaunchfailure", "device-side assert")):
                raise
            ok = False
        if not (t == t and t > 0.0):        # NaN or not a duration
            ok = False
        if agree is not None:
            ok = bool(agree(ok))
        if ok:
            times[name] = t
    if not times:
        raise RuntimeError("no launch mode could be measured (candidates: %s)%s" % (
            ", ".join(candidates), "; " + "; ".join("%s -> %s" % kv for kv in (errors or {}).items()) if errors else ""))
    best = min(times, key=lambda k: (times[k], list(candidates).index(k)))
    return best, times


def seed_rank_rng(base_seed, rank):
    """Distinct numpy streams per rank: the Infomax negative-sampling permutation
    (graphcnn.py:199) is drawn over the LOCAL batch."""
    np.random.seed(base_seed + 7919 * rank)
