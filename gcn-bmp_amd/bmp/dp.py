"""Data-parallel training step plumbing: one flat fp32 parameter/gradient buffer, ONE RCCL
all-reduce per step, Adam with Chainer's update rule.

Replaces the reference's in-process 2-GPU ``ParallelUpdater`` (train_binary.py:546-549): one
process per GPU, drug pairs sharded by rank, gradients summed with a single ``all_reduce`` over
xGMI (the whole model is 0.3-3 M floats, so the collective is latency-bound and needs no
bucketing or overlap).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.distributed as dist
from torch import nn
from torch.autograd import Function


class _Unflatten(Function):
    """flat parameter buffer -> the individual parameter tensors (views).  The backward writes all
    parameter gradients into ONE flat tensor with a single concatenation, instead of one
    accumulate kernel per parameter."""

    @staticmethod
    def forward(ctx, flat, shapes):
        ctx.shapes = shapes
        outs, off = [], 0
        for shp in shapes:
            n = 1
            for k in shp:
                n *= k
            outs.append(flat[off:off + n].view(shp))
            off += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        parts = []
        for g, shp in zip(grads, ctx.shapes):
            if g is None:
                n = 1
                for k in shp:
                    n *= k
                parts.append(grads[0].new_zeros(n) if grads[0] is not None else torch.zeros(n))
            else:
                parts.append(g.reshape(-1))
        return torch.cat(parts), None


class FlatAdam:
    """Flattens a module's parameters and gradients into two contiguous buffers (the parameters
    become views) and applies chainer.optimizers.Adam (train_ddi_modify.py:289):
    alpha_t = alpha*sqrt(1-b2^t)/(1-b1^t);  p -= alpha_t*m/(sqrt(v)+eps) + weight_decay_rate*p."""

    def __init__(self, module: nn.Module, alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay_rate=0.0,
                 process_group: Optional["dist.ProcessGroup"] = None):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        total = sum(p.numel() for p in params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            p.grad = self.grad[off:off + n].view_as(p)
            off += n
        self.params = params
        self.names = [n for n, p in module.named_parameters() if p.requires_grad]
        self.shapes = [tuple(p.shape) for p in params]
        self.module = module
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.t = 0
        self.alpha, self.beta1, self.beta2, self.eps, self.wd = alpha, beta1, beta2, eps, weight_decay_rate
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1

    def zero_grad(self) -> None:
        self.grad.zero_()

    # ---- functional mode: one gradient tensor for the whole model, no per-parameter accumulation ----
    def functional_forward(self, *args, **kwargs):
        """Run the module on parameter views produced by ONE autograd node over the flat buffer.
        After ``loss.backward()`` call ``collect_grads()``; the flat gradient is then in ``self.grad``."""
        leaf = self.flat.detach().requires_grad_()
        self._leaf = leaf
        views = _Unflatten.apply(leaf, self.shapes)
        return torch.func.functional_call(self.module, dict(zip(self.names, views)), args, kwargs)

    def collect_grads(self) -> None:
        g = self._leaf.grad
        self.grad = g if g is not None else torch.zeros_like(self.flat)
        self._leaf = None

    def reattach(self) -> None:
        """Fold any .grad tensor that autograd (or a caller) replaced back into the flat buffer."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad[off:off + n].data_ptr():
                if p.grad is not None:
                    self.grad[off:off + n].add_(p.grad.reshape(-1))
                p.grad = self.grad[off:off + n].view_as(p)
            off += n

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world > 1:
            dist.broadcast(self.flat, src=src, group=self.group)

    def all_reduce_grads(self) -> None:
        """The step's single collective: sum over ranks, then the mean (each rank's loss is the
        mean over its own shard)."""
        if self.world > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)
            self.grad.mul_(1.0 / self.world)

    def step(self) -> None:
        self.t += 1
        a_t = self.alpha * math.sqrt(1.0 - self.beta2 ** self.t) / (1.0 - self.beta1 ** self.t)
        g = self.grad
        self.m.mul_(self.beta1).add_(g, alpha=1.0 - self.beta1)
        self.v.mul_(self.beta2).addcmul_(g, g, value=1.0 - self.beta2)
        if self.wd:
            self.flat.mul_(1.0 - self.wd)
        self.flat.addcdiv_(self.m, self.v.sqrt().add_(self.eps), value=-a_t)


def shard(n_items: int, rank: int, world: int):
    """Rank r takes items [r*n/W, (r+1)*n/W) of every global batch."""
    per = n_items // world
    return slice(rank * per, (rank + 1) * per)
