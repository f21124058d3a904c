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
        ctx.set_materialize_grads(False)          # unused parameters arrive as None, not as one zero-fill launch each
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
        some = next((g for g in grads if g is not None), None)
        if some is None:
            return None, None
        sizes = []
        for shp in ctx.shapes:
            n = 1
            for k in shp:
                n *= k
            sizes.append(n)
        if all(g is not None for g in grads):
            return torch.cat([g.reshape(-1) for g in grads]), None
        flat = some.new_zeros(sum(sizes))         # parameters the layout plan manages arrive as None: one fill,
        off = 0                                   # then only the autograd-managed slices are copied in
        for g, n in zip(grads, sizes):
            if g is not None:
                flat[off:off + n].copy_(g.reshape(-1))
            off += n
        return flat, None


class FlatAdam:
    """Flattens a module's parameters and gradients into two contiguous buffers (the parameters
    become views) and applies chainer.optimizers.Adam (train_ddi_modify.py:289):
    alpha_t = alpha*sqrt(1-b2^t)/(1-b1^t);  p -= alpha_t*m/(sqrt(v)+eps) + weight_decay_rate*p."""

    def __init__(self, module: nn.Module, alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay_rate=0.0,
                 process_group: Optional["dist.ProcessGroup"] = None):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no parameters")
        if any(isinstance(p, nn.UninitializedParameter) for p in params):
            raise ValueError("the module still has a lazily sized Linear (MLP / SymMLP / HolE built without in_dim / fp_dim): "
                             "call it once, or build it inside GraphConvPredictorForPair, before FlatAdam flattens the parameters")
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
        self.plan = None
        self._plan_tried = False
        self._gscale = 1.0

    def zero_grad(self) -> None:
        self.grad.zero_()

    # ---- functional mode: one gradient tensor for the whole model, no per-parameter accumulation ----
    def functional_forward(self, *args, **kwargs):
        """Run the module on parameter views produced by ONE autograd node over the flat buffer.
        After ``loss.backward()`` call ``collect_grads()``; the flat gradient is then in ``self.grad``."""
        return self._functional(self.module, args, kwargs)

    def _functional(self, call, args, kwargs):
        leaf = self.flat.detach().requires_grad_()
        self._leaf = leaf
        views = _Unflatten.apply(leaf, self.shapes)
        plan = self._layout_plan()
        hooked = []
        if plan is not None:
            # one launch: every kernel-layout weight array of the planned sub-modules; their weight gradients stay in
            # the plan's buffers until collect_grads()
            plan.prepare(self.flat)
            tape = leaf[:1]
            for prefix, mod in self._plan_sections:
                object.__setattr__(mod, "_fast", (plan.P[prefix], plan.G[prefix], plan.state, tape))   # (nn.Module.__setattr__: 6 us)
                hooked.append(mod)
        # The views stand in for the parameters during the call -- what torch.func.functional_call does, without its
        # per-call walk over the module tree (0.3 ms of host time per step; the RelGCN step is 1.5 ms): every registration
        # of every parameter (tied ones included) was located once.
        slots = self._param_slots()
        try:
            for (reg, key, _orig), k in slots:
                reg[key] = views[k]
            return call(*args, **kwargs)
        finally:
            for (reg, key, orig), _k in slots:
                reg[key] = orig
            for mod in hooked:
                object.__setattr__(mod, "_fast", None)

    def functional_loss(self, *args, t):
        """``functional_forward`` through the module's ``forward_loss`` (GraphConvPredictorForPair: the reference's Classifier,
        train_ddi_modify.py:284-286): the loss of the batch against the labels ``t``, with link predictor + loss as one launch
        each way where the module offers it.  Then ``loss.backward()``, ``collect_grads()``, ``step()`` as usual."""
        return self._functional(lambda *a: self.module.forward_loss(*a, t=t), args, {})

    def functional_predict(self, *args, **kwargs):
        """The evaluation callers' ``predict`` (eval_coattention.py:103-124; training/extensions/batch_evaluator.py:49-100 runs it
        over the train and validation sets every epoch) on the planned path: logits and the two molecule vectors handed to the
        link predictor, under no-backprop.  The kernels then keep nothing for a backward (no m / r|z / c / ij / C stores)."""
        with torch.no_grad():
            y = self.functional_forward(*args, **kwargs)
        return y, (getattr(self.module, "g1", None), getattr(self.module, "g2", None))

    def _param_slots(self):
        """[((module._parameters, name, parameter), index into self.params)] over every registration of a flattened
        parameter in the module tree."""
        if getattr(self, "_slots", None) is None:
            index = {id(p): k for k, p in enumerate(self.params)}
            slots = []
            for mod in self.module.modules():
                for key, prm in mod._parameters.items():
                    if prm is not None and id(prm) in index:
                        slots.append(((mod._parameters, key, prm), index[id(prm)]))
            self._slots = slots
        return self._slots

    def _layout_plan(self):
        """bmp.plan.LayoutPlan over the sub-modules that support it (GGNN encoder, fine co-attention), built at the
        first functional step on a GPU."""
        if self._plan_tried or not self.flat.is_cuda:
            return self.plan
        self._plan_tried = True
        from .plan import LayoutPlan
        sections = []
        for name, mod in self.module.named_modules():
            if name and callable(getattr(mod, "plannable", None)) and mod.plannable() and hasattr(mod, "prepared_layouts"):
                if all(p.requires_grad for p in mod.parameters()):
                    sections.append((name + ".", mod))
        if sections:
            self._plan_sections = sections
            self.plan = LayoutPlan(sections, self.names, self.shapes, self.flat.device)
        return self.plan

    def collect_grads(self) -> None:
        g = self._leaf.grad
        self._leaf = None
        if self.plan is not None:
            if self.plan.state.pop("head_gscale", None) is not None:
                # MLPLossFn.backward handed the factor arriving at the loss to the co-attention node, whose backward did not
                # run (backward(inputs=...) past it, a frozen co-attention): the rows' gradients went on unscaled
                raise RuntimeError("the loss-gradient factor left for the co-attention's backward was never consumed; "
                                   "call loss.backward() through the whole model or use functional_forward + model.loss")
            # one launch: (+)= the planned modules' parameter gradients; when autograd carried none (every parameter is the
            # plan's), that launch writes the whole buffer and no zero-fill precedes it
            self.grad = g if g is not None else torch.empty_like(self.flat)
            self.plan.collect(self.grad, overwrite=g is None)
        else:
            self.grad = g if g is not None else torch.zeros_like(self.flat)

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
            if self.grad.is_cuda:
                self._gscale = 1.0 / self.world      # folded into the Adam kernel
            else:
                self.grad.mul_(1.0 / self.world)

    def step(self) -> None:
        self.t += 1
        a_t = self.alpha * math.sqrt(1.0 - self.beta2 ** self.t) / (1.0 - self.beta1 ** self.t)
        g = self.grad
        if self.flat.is_cuda:                        # the whole update rule as one kernel over the flat buffers
            from . import _lib
            from ._lib import check, ptr, stream
            a_dev = getattr(self, "_alpha_dev", None)        # set by GraphedTrainStep: alpha_t lives on the device there
            check(_lib.lib().bmp_adam_step(ptr(self.flat), ptr(g), ptr(self.m), ptr(self.v), self.flat.numel(), a_t,
                                           ptr(a_dev), self.beta1, self.beta2, self.eps, self.wd, self._gscale, stream()),
                  "bmp_adam_step")
            self._gscale = 1.0
            return
        self.m.mul_(self.beta1).add_(g, alpha=1.0 - self.beta1)
        self.v.mul_(self.beta2).addcmul_(g, g, value=1.0 - self.beta2)
        if self.wd:
            self.flat.mul_(1.0 - self.wd)
        self.flat.addcdiv_(self.m, self.v.sqrt().add_(self.eps), value=-a_t)


def shard(n_items: int, rank: int, world: int):
    """Rank r takes items [r*n/W, (r+1)*n/W) of every global batch."""
    per = n_items // world
    return slice(rank * per, (rank + 1) * per)


class GraphedTrainStep:
    """One whole training step (forward, loss, backward, gradient all-reduce, Adam) recorded once per distinct batch -- or
    once for ALL batches of a size, when they are loaded into one ``bmp.packed.StaticPairBatch`` (round 4) -- as a HIP graph
    and replayed: ~55 launches become one submission and the host's per-launch work leaves the step.  Measured: at the reference's default batch of 32
    pairs (train_ddi_modify.py:196) 1.64 -> 1.57 ms per step -- that step is a chain of ~70 dependent kernels of one
    or two workgroup rounds each, bound by their latencies rather than by the host; at 1024 pairs per GPU the step is
    GPU-bound and gains nothing.  Kept as the host-jitter-free way to run a fixed set of batches.

    Everything a replay depends on must sit at fixed device addresses: the packed batch and its labels (device
    resident already), the flat parameter / moment buffers, and Adam's step-dependent factor alpha_t, which is read
    from a one-element device tensor updated before every replay.  Host-side decisions made while recording (which
    gradient buffer a kernel overwrites, which it accumulates into) are the same for every replay of one batch.
    """

    def __init__(self, model, opt: "FlatAdam", warmup: int = 2):
        if not opt.flat.is_cuda:
            raise ValueError("graphs need a GPU")
        self.model, self.opt, self.warmup = model, opt, warmup
        self.graphs = {}
        self._alpha = torch.zeros(1, dtype=torch.float32, device=opt.flat.device)

    def _set_alpha(self) -> None:
        o = self.opt
        self._alpha.fill_(o.alpha * math.sqrt(1.0 - o.beta2 ** (o.t + 1)) / (1.0 - o.beta1 ** (o.t + 1)))

    def _body(self, pb, t, static=None):
        plan = self.opt._layout_plan()
        if plan is None:
            return self._body_streams(pb, t, static)
        was = getattr(plan, "in_line", False)
        plan.in_line = True                 # one in-order chain: what a replay runs back to back (bmp/plan.py: prepare)
        try:
            return self._body_streams(pb, t, static)
        finally:
            plan.in_line = was

    def _body_streams(self, pb, t, static=None):
        o = self.opt
        if static is not None:
            # a batch at fixed addresses (bmp.packed.StaticPairBatch): its arrays are written by the first launch of the step,
            # and what an earlier step derived from their contents is derived again
            static.reset_derived()
            static.emit()
        if callable(getattr(self.model, "forward_loss", None)):      # the reference's Classifier: link predictor + loss together
            loss = o.functional_loss(pb, t=t)
        else:
            loss = self.model.loss(o.functional_forward(pb), t)
        loss.backward()
        o.collect_grads()
        o.all_reduce_grads()
        o._alpha_dev = self._alpha          # Adam reads alpha_t from the device in the recorded step (and only there: eager
        try:                                # steps between replays take it from the host as ever)
            o.step()
        finally:
            o._alpha_dev = None
        return loss

    def __call__(self, pb, t=None) -> torch.Tensor:
        """One step on the batch ``(pb, t)`` -- or on a ``StaticPairBatch`` (``t`` is then its own label array): ONE graph
        serves every batch loaded into it."""
        static = pb if callable(getattr(pb, "emit", None)) else None
        if static is not None:
            if self.opt.world > 1:
                # (one rank per GPU records its own step; the all-reduce inside a recording has never run on this stack)
                raise NotImplementedError("a step recorded on a fixed-shape batch is a single-rank path here: the RCCL all-reduce "
                                          "inside a HIP graph is untested; use the packed layouts for N > 1")
            pb, t = static.pb, static.t
        key = (id(pb), id(t))
        o = self.opt
        if key not in self.graphs:
            # warm-up on a side stream (allocator, lazy builds), with the optimizer state put back afterwards
            saved = (o.flat.clone(), o.m.clone(), o.v.clone(), o.t)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(self.warmup):
                    self._set_alpha()
                    self._body(pb, t, static)
            torch.cuda.current_stream().wait_stream(s)
            o.flat.copy_(saved[0]); o.m.copy_(saved[1]); o.v.copy_(saved[2]); o.t = saved[3]
            g = torch.cuda.CUDAGraph()
            t_before = o.t
            with torch.cuda.graph(g):
                loss = self._body(pb, t, static)
            o.t = t_before                      # recording does not execute: the step count advances on replay
            self.graphs[key] = (g, loss, pb, t)
        g, loss, _pb, _t = self.graphs[key]
        self._set_alpha()
        g.replay()
        o.t += 1
        return loss
