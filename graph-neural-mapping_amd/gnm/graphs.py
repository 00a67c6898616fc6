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
import torch

from .arena import StaticBatch


class CapturedTrainStep:
    def __init__(self, model, template_batch, loss_fn, zero_grad=None, warmup=3, post_backward=None, preserve=(),
                 agg0_cache=True):
        """loss_fn(c_logit, d_logit, labels) -> scalar loss.  zero_grad(): clears the
        gradient buffers (default: model.zero_grad(set_to_none=False)).  post_backward(): extra
        capturable work recorded after backward (e.g. a fused optimizer step).  The warm-up passes
        are real steps on the template batch: parameters, BatchNorm buffers and every tensor in
        `preserve` (optimizer state) are put back afterwards, so construction has no side effect."""
        self.model = model
        self._post = post_backward
        self._agg0_cache = agg0_cache      # False: aggregate the input features inside the step (bench.py)
        keep = [t for t in model.state_dict().values()] + list(preserve)
        snapshot = [t.clone() for t in keep]
        dev = template_batch.node_off.device
        self.static = StaticBatch(template_batch)
        B = template_batch.B
        self.labels = torch.zeros(B, dtype=torch.int64, device=dev)
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
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = self._step()
        with torch.no_grad():
            for t, s0 in zip(keep, snapshot):
                t.copy_(s0)
        torch.cuda.synchronize()
        self._arena_ptrs = self._arena_buffers()

    def _arena_buffers(self):
        """addresses the captured kernels read the graph pool from: adding graphs may re-allocate them"""
        a = self.static.batch.arena
        return (a.rowptr.buf.data_ptr(), a.col.buf.data_ptr(), a.feat.buf.data_ptr(),
                tuple(sorted((k, v["buf"].data_ptr()) for k, v in a._agg0.items() if v["buf"] is not None)))

    def _step(self):
        self._zero()
        bt = self.static.batch
        X = None if self._agg0_cache else bt.arena.features(bt)
        c_logit, d_logit = self.model.forward_batch(bt, X=X, perm=self.perm)
        loss = self._loss_fn(c_logit, d_logit, self.labels)
        loss.backward()
        if self._post is not None:
            self._post()
        return loss

    def run(self, batch, labels, perm=None):
        """Replay the step on `batch` (same shape as the template).  perm: the
        np.random.permutation(B) of graphcnn.py:199 (drawn here if None)."""
        if perm is None:
            perm = np.random.permutation(batch.B)
        if self._arena_buffers() != self._arena_ptrs:
            raise RuntimeError("the graph arena was re-allocated after this step was captured (graphs were added): "
                               "add every graph before building CapturedTrainStep / FusedTrainStep, or build a new one")
        self.static.load(batch)
        self.labels.copy_(labels, non_blocking=True)
        self.perm.copy_(torch.as_tensor(np.asarray(perm), dtype=torch.int32).pin_memory(), non_blocking=True)
        self.graph.replay()
        return self.loss
