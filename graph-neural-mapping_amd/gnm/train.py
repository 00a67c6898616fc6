"""Fused train-step tail: the losses of the reference's train() and its Adam/StepLR as single
launches on flat buffers, and a whole-step driver around them.

What the reference does per step (main.py:29-41) and what runs here instead:

    c_loss = CrossEntropyLoss()(c_logit, c_labels)             \\
    d_loss = BCEWithLogitsLoss()(d_logit, d_labels)              >  infomax_loss(): gnm_loss_ce_bce (2 launches) and,
    loss = c_loss + beta * d_loss                               /   in backward, gnm_loss_ce_bce_grad (1 launch)
    optimizer.zero_grad(); loss.backward()                      -> the model's hand-written backward; gradients
                                                                   land in the flat buffer (GinSpec.grad_sink)
    optimizer.step()            # Adam over ~30 tensors         -> FusedAdam.step(): gnm_adam_step, one launch
    scheduler.step()            # StepLR, main.py:137,153       -> StepLR.step(): an 8-byte write to the device

Everything is hipGraph-capturable (step count and learning rate live in device memory), so
FusedTrainStep can replay forward + loss + backward + Adam as one graph; with more than one rank
the gradient all-reduce sits between the captured part and the Adam launch.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._cabi import check, lib
from .core import _stream, _stream_scope
from .parallel import DataParallelGIN


class _InfomaxLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c_logit, d_logit, labels, d_target, beta, n_pos, hold=None):
        if not c_logit.is_cuda:
            raise RuntimeError("infomax_loss runs on the GPU only (libgnm_hip.so)")
        # (`parts` is not differentiable: without this autograd hands backward() a freshly zero-filled gradient for it --
        #  a fill launch per step)
        ctx.set_materialize_grads(False)
        c = c_logit.contiguous()
        d = d_logit.contiguous().view(-1)
        B, C_ = c.shape
        M = d.numel()
        dev = c.device
        loss3 = torch.empty(3, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.gnm_loss_workspace_doubles(M)), dtype=torch.float64, device=dev)
        tgt = None
        if d_target is not None:
            tgt = d_target.to(torch.float32).contiguous().view(-1)
            if tgt.numel() != M:
                raise ValueError("d_labels has %d entries, d_logit %d" % (tgt.numel(), M))
        lab = labels.to(torch.int64).contiguous()
        if lab.numel() != B:
            raise ValueError("labels has %d entries, c_logit %d rows" % (lab.numel(), B))
        with _stream_scope(dev):
            check(lib.gnm_loss_ce_bce(c.data_ptr(), c.stride(0), lab.data_ptr(), B, C_, d.data_ptr(),
                                      tgt.data_ptr() if tgt is not None else None, M, int(n_pos), float(beta),
                                      loss3.data_ptr(), None, 0, None, ws.data_ptr(), _stream()), "gnm_loss_ce_bce")
        # hold: the model's DiscUnit hand-over (gnm/core.py), only with the reference's default targets
        ctx.hold = hold if (tgt is None and 2 * int(n_pos) == M and M > 0) else None
        ctx.args = (c, d, lab, tgt, int(n_pos), float(beta), d_logit.shape)
        parts = loss3.detach()
        ctx.mark_non_differentiable(parts)
        return loss3[0], parts

    @staticmethod
    def backward(ctx, g, _gparts):
        c, d, lab, tgt, n_pos, beta, dshape = ctx.args
        if g is None:                  # only `parts` was used downstream: nothing flows back
            return (None,) * 7
        B, C_ = c.shape
        M = d.numel()
        g = g.to(torch.float32).contiguous()
        dC, dD = torch.empty_like(c), torch.empty_like(d)
        with _stream_scope(c.device):      # both gradients, already multiplied by the upstream gradient, in one launch
            check(lib.gnm_loss_ce_bce_grad(c.data_ptr(), c.stride(0), lab.data_ptr(), B, C_, d.data_ptr(),
                                           tgt.data_ptr() if tgt is not None else None, M, n_pos, beta, g.data_ptr(),
                                           dC.data_ptr(), dC.stride(0), dD.data_ptr(), _stream()),
                  "gnm_loss_ce_bce_grad")
        dD = dD.view(dshape)
        if ctx.hold is not None:
            # dD = k (sigmoid(d_logit) - target) with the kernel's fp32 factor k = upstream * (beta / M): the model's
            # backward scales the reductions its forward left instead of re-reading the hidden layers
            ctx.hold.k = g.reshape(1)
            ctx.hold.kscale = float(np.float32(beta) / np.float32(M))
            ctx.hold.dD_ptr = dD.data_ptr()
            ctx.hold.dD_version = dD._version      # an in-place gradient hook (g.clamp_(), g.mul_()) keeps the pointer
        return dC, dD, None, None, None, None, None


def infomax_loss(c_logit, d_logit, labels, beta=0.05, d_labels=None):
    """loss = CE(c_logit, labels) + beta * BCEWithLogits(d_logit, d_labels) (main.py:34-37).
    d_labels None = the reference's targets: ones for the first half of d_logit (true pairs), zeros for the
    shuffled half (main.py:32 with the node count).  Returns (loss, parts) with parts = [loss, c_loss, d_loss]
    (detached, on the device: reading them is the caller's sync, as main.py:43 does)."""
    n_pos = d_logit.numel() // 2
    hold = getattr(d_logit, "_gnm_disc_unit", None) if d_labels is None else None
    return _InfomaxLossFn.apply(c_logit, d_logit, labels, d_labels, beta, n_pos, hold)


class FusedAdam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) semantics on the flat parameter / gradient
    buffers of gnm.parallel.FlatParams; one launch per step."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.fp = flat
        dev = flat.flat.device
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self._hyper_host = [float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), 1.0]
        self.hyper = torch.tensor(self._hyper_host, dtype=torch.float64, device=dev)
        self.param_groups = [{"lr": float(lr), "initial_lr": float(lr)}]      # what LR schedulers look at

    @property
    def lr(self):
        return self._hyper_host[0]

    def _push(self):
        self.hyper.copy_(torch.tensor(self._hyper_host, dtype=torch.float64), non_blocking=True)

    def set_lr(self, lr):
        self._hyper_host[0] = float(lr)
        self.param_groups[0]["lr"] = float(lr)
        self._push()

    def set_grad_scale(self, s):
        """gradients are multiplied by s before use (1/world after a SUM all-reduce)"""
        self._hyper_host[5] = float(s)
        self._push()

    def zero_grad(self, set_to_none=False):
        self.fp.zero_grad()

    def step(self):
        with _stream_scope(self.fp.flat.device):
            check(lib.gnm_adam_step(self.fp.flat.data_ptr(), self.fp.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                    self.exp_avg_sq.data_ptr(), self.fp.total, self.hyper.data_ptr(),
                                    self.step_count.data_ptr(), _stream()), "gnm_adam_step")

    def state_dict(self):
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "step": self.step_count.clone(), "hyper": list(self._hyper_host)}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count.copy_(sd["step"])
        self._hyper_host = list(sd["hyper"])
        self.param_groups[0]["lr"] = self._hyper_host[0]
        self._push()


class StepLR:
    """optim.lr_scheduler.StepLR(optimizer, step_size, gamma) (main.py:137): lr = lr0 * gamma^(epoch // step_size)."""

    def __init__(self, optimizer, step_size, gamma=0.1):
        self.opt, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.base_lr = optimizer.lr
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        self.opt.set_lr(self.base_lr * self.gamma ** (self.last_epoch // self.step_size))

    def get_last_lr(self):
        return [self.opt.lr]


class FusedTrainStep:
    """One object = the body of the reference's train() loop (main.py:24-41) for a GIN_InfoMaxReg.

        step = FusedTrainStep(model, lr=0.01, beta=0.05, template_batch=arena.batch_from_gids(gids))
        parts = step.run(batch, labels)            # [loss, c_loss, d_loss] on the device, no sync

    capture=True (needs template_batch; every later batch must have its shape) records forward + loss +
    backward (+ Adam when single-rank) into a hipGraph.  With a process group the flat gradient buffer is
    SUM-all-reduced after the captured part and Adam applies the 1/world factor.
    template_gids (the template batch's arena ids on the host, equal-size graphs): enables run_gids(ids, labels) --
    the data-loader route: ids, labels and permutation reach the device through one pinned upload (gnm/graphs.py)."""

    def __init__(self, model, lr=1e-3, beta=0.05, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 process_group=None, template_batch=None, capture=True, template_gids=None):
        self.model, self.beta = model, float(beta)
        self.dp = DataParallelGIN(model, process_group)
        if not self.dp.direct:
            raise RuntimeError("FusedTrainStep needs the sum/average neighbour-pooling model (gradient sink)")
        self.dp.broadcast_parameters()          # every replica starts from rank 0's parameters and buffers
        self.optimizer = FusedAdam(self.dp.fp, lr, betas, eps, weight_decay)
        self.optimizer.set_grad_scale(1.0 / self.dp.world)
        self.parts = None
        self.captured = None
        if capture:
            if template_batch is None:
                raise ValueError("capture=True needs a template_batch")
            from .graphs import CapturedTrainStep
            post = self.optimizer.step if self.dp.world == 1 else None
            opt = self.optimizer
            self.captured = CapturedTrainStep(model, template_batch, self._loss, zero_grad=self.dp.zero_grad,
                                              post_backward=post,
                                              preserve=(opt.exp_avg, opt.exp_avg_sq, opt.step_count),
                                              gids_host=template_gids)

    def _loss(self, c_logit, d_logit, labels):
        loss, self.parts = infomax_loss(c_logit, d_logit, labels, self.beta)
        return loss

    def run(self, batch, labels, perm=None):
        if self.captured is not None:
            self.captured.run(batch, labels, perm)
            if self.dp.world > 1:
                self._allreduce_sum()
                self.optimizer.step()
        else:
            if perm is None:
                perm = np.random.permutation(batch.B)
            self.dp.zero_grad()
            c_logit, d_logit = self.model.forward_batch(batch, perm=perm)
            self._loss(c_logit, d_logit, labels).backward()
            self._allreduce_sum()
            self.optimizer.step()
        return self.parts

    def run_gids(self, gids_host, labels_host, perm=None):
        """run() for a selection given as arena ids + labels on the host (needs template_gids at construction)"""
        if self.captured is None or self.captured.packed is None:
            raise RuntimeError("run_gids needs FusedTrainStep(..., capture=True, template_gids=...)")
        self.captured.run_gids(gids_host, labels_host, perm)
        if self.dp.world > 1:
            self._allreduce_sum()
            self.optimizer.step()
        return self.parts

    def _allreduce_sum(self):
        if self.dp.world > 1:
            dist.all_reduce(self.dp.fp.flat_grad, op=dist.ReduceOp.SUM, group=self.dp.group)
