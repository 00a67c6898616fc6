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
    def __init__(self, model, template_batch, loss_fn, zero_grad=None, warmup=3):
        """loss_fn(c_logit, d_logit, labels) -> scalar loss.  zero_grad(): clears the
        gradient buffers (default: model.zero_grad(set_to_none=False))."""
        self.model = model
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
        torch.cuda.synchronize()

    def _step(self):
        self._zero()
        c_logit, d_logit = self.model.forward_batch(self.static.batch, perm=self.perm)
        loss = self._loss_fn(c_logit, d_logit, self.labels)
        loss.backward()
        return loss

    def run(self, batch, labels, perm=None):
        """Replay the step on `batch` (same shape as the template).  perm: the
        np.random.permutation(B) of graphcnn.py:199 (drawn here if None)."""
        if perm is None:
            perm = np.random.permutation(batch.B)
        self.static.load(batch)
        self.labels.copy_(labels, non_blocking=True)
        self.perm.copy_(torch.as_tensor(np.asarray(perm), dtype=torch.int32), non_blocking=True)
        self.graph.replay()
        return self.loss
