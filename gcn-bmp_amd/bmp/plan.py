"""Layout plan: the host glue of a training step as two table-driven kernel launches.

The kernels want weights in their own layouts (K-major transposes, K4 packs, the folded GRU matrices,
concatenated readout / co-attention operands: ``prepared_layouts()`` of the modules), and they emit weight
gradients in yet another set of buffers (``gk``: o1 / o2 / dUcT / cs per step group, dWbT, ...).  In eager
mode those conversions are ~100 tiny framework kernels and autograd nodes per step.  Every one of them is a
fixed linear map with 0/1 coefficients, so the plan evaluates the modules' own layout code ONCE, on the CPU,
on index-valued stand-ins for the parameters (element j of the flat parameter buffer carries the value j+1),
and reads off two gather tables:

* ``prepare``:  prep[i]      = sum_k flat[tab_p[k][i]]        (all kernel-layout arrays, one launch)
* ``collect``:  flat_grad[j] += sum_k gk[tab_g[k][j]]         (all parameter gradients, one launch)

Because the tables come from the same functions the eager path runs (and the eager path is checked against
the oracle), the two paths cannot drift apart; tests/test_plan.py checks them against each other.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import check, ptr, stream


class _StandIns:
    """Temporarily replaces a module's parameters by CPU float64 tensors of the same shapes (the module may hold
    non-leaf tensors from an earlier forward, so it cannot be deep-copied)."""

    def __init__(self, mod: nn.Module):
        self.mod = mod
        self.slots = []
        for name, p in mod.named_parameters():
            owner = mod
            parts = name.split(".")
            for a in parts[:-1]:
                owner = getattr(owner, a)
            self.slots.append((owner, parts[-1], p, torch.zeros(tuple(p.shape), dtype=torch.float64)))

    def __enter__(self):
        for owner, key, _p, t in self.slots:
            owner._parameters[key] = t
        return [t for _o, _k, _p, t in self.slots]

    def __exit__(self, *exc):
        for owner, key, p, _t in self.slots:
            owner._parameters[key] = p
        return False


def _index_eval(mod: nn.Module, fn_name: str, params: List[torch.Tensor]) -> Dict[str, List[np.ndarray]]:
    """Evaluate ``mod.<fn_name>()`` once per parameter tensor with that tensor index-valued (local element e ->
    e + 1) and all others zero (``params``: the module's float64 stand-ins).  Returns name -> list over
    parameters of int64 arrays (0 = no contribution, else 1 + local element index)."""
    out: Dict[str, List[np.ndarray]] = {}
    with torch.no_grad():
        for p in params:
            p.zero_()
        for k, p in enumerate(params):
            p.copy_(torch.arange(1, p.numel() + 1, dtype=p.dtype).view_as(p))
            res = getattr(mod, fn_name)()
            for name, t in res.items():
                a = t.detach().reshape(-1).numpy()
                ai = np.rint(a).astype(np.int64)
                if not np.array_equal(ai.astype(a.dtype), a):
                    raise AssertionError(f"layout {name} is not a 0/1 map of parameter #{k}")
                out.setdefault(name, [None] * len(params))[k] = ai
            p.zero_()
    return out


class SideStream:
    """A lowest-priority HIP stream (bmp_stream_create_low) with a workspace of its own, for the launches
    functional._on_side puts beside the backward chain."""

    def __init__(self, device):
        import ctypes
        device = torch.device(device)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(_lib.lib().bmp_stream_create_low(ctypes.byref(h)), "bmp_stream_create_low")
        self.handle = h
        self.stream = torch.cuda.ExternalStream(h.value, device=device)
        self.ws = None
        self.keep: list = []          # what the enqueued launches read: released once the streams have joined

    def workspace(self, nfloats: int, device) -> torch.Tensor:
        if self.ws is None or self.ws.numel() < nfloats:
            if self.ws is not None:
                self.keep.append(self.ws)
            self.ws = torch.empty(max(int(nfloats), 4), dtype=torch.float32, device=device)
            self.stream.wait_stream(torch.cuda.current_stream())       # the block's earlier users are on that stream
        return self.ws

    def join(self) -> None:
        """The current stream waits for everything enqueued here; the held tensors go back to the allocator (whatever
        reuses them is enqueued behind the wait)."""
        torch.cuda.current_stream().wait_stream(self.stream)
        self.keep.clear()

    def __del__(self):
        try:
            self.stream.synchronize()
            _lib.lib().bmp_stream_destroy(self.handle)
        except Exception:
            pass


class PartStream:
    """An ordinary stream for the second half of the planned encoder's tile-local forward launches
    (functional._fwd_parts).  ``extra`` more streams (BMP_FWD_CHAINS - 2, default none): the four-chain diagnostic form of
    DESIGN.md 3a' (the stream-count cliff); the product runs two chains."""

    def __init__(self, device, extra: int = 0):
        import ctypes
        self.stream = torch.cuda.Stream(device=torch.device(device))
        self.handle = ctypes.c_void_p(self.stream.cuda_stream)
        self.more = [torch.cuda.Stream(device=torch.device(device)) for _ in range(max(int(extra), 0))]
        self.more_handles = [ctypes.c_void_p(s.cuda_stream) for s in self.more]

    def join(self) -> None:
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.stream)
        for s in self.more:
            cur.wait_stream(s)


_SHARED_STREAMS: Dict[tuple, object] = {}


def _shared(kind: str, device, make):
    d = torch.device(device)
    key = (kind, d.type, d.index if d.index is not None else torch.cuda.current_device())
    if key not in _SHARED_STREAMS:
        _SHARED_STREAMS[key] = make()
    return _SHARED_STREAMS[key]


class LayoutPlan:
    """Built for a list of (prefix, module) pairs whose parameters are slices of one flat fp32 buffer.
    A module takes part if it defines ``prepared_layouts()``, ``primary_layouts()``, ``gk_spec()`` and
    ``primary_grads(gk)`` (bmp.ggnn.GGNN, bmp.coattention.NieFineCoattention / PoolingFineCoattention)."""

    def __init__(self, sections: List[Tuple[str, nn.Module]], names: List[str], shapes: List[tuple], device):
        self.device = device
        n_flat = int(sum(int(np.prod(s)) for s in shapes))
        off_of = {}
        o = 0
        for n, s in zip(names, shapes):
            off_of[n] = o
            o += int(np.prod(s))
        self.managed = set()
        self.prep_slices: Dict[str, Dict[str, Tuple[int, tuple]]] = {}
        self.gk_slices: Dict[str, Dict[str, Tuple[int, tuple]]] = {}
        prep_off, gk_off = 0, 0
        grad_lists: Dict[int, List[int]] = {}                  # flat param element -> gk positions
        tabs_p: List[np.ndarray] = []
        for prefix, mod in sections:
            pnames = [prefix + n for n, _ in mod.named_parameters()]
            for n in pnames:
                if n not in off_of:
                    raise KeyError(f"parameter {n} is not in the flat buffer")
                self.managed.add(n)
            poffs = [off_of[n] for n in pnames]
            with _StandIns(mod) as stand:
                prep_off, gk_off = self._section(prefix, mod, stand, poffs, prep_off, gk_off, tabs_p, grad_lists)
        self.n_prep, self.n_gk, self.n_flat = prep_off, gk_off, n_flat
        tab_p = _concat_tables(tabs_p)
        Kg = max((len(v) for v in grad_lists.values()), default=1)
        tab_g = np.full((Kg, n_flat), -1, dtype=np.int32)
        for j, qs in grad_lists.items():
            tab_g[:len(qs), j] = sorted(qs)
        self.Kp, self.Kg = tab_p.shape[0], Kg
        self.tab_p_host, self.tab_g_host = tab_p, tab_g
        self.tab_p = torch.from_numpy(tab_p).to(device)
        self.tab_g = torch.from_numpy(tab_g).to(device)
        self.prep = torch.empty(max(prep_off, 1), dtype=torch.float32, device=device)
        self.gk = torch.zeros(max(gk_off, 1), dtype=torch.float32, device=device)
        # weight-gradient launches beside the backward chain (functional._on_side); BMP_WGRAD_STREAM=0 keeps them in line
        one = os.environ.get("BMP_ONE_STREAM") == "1"        # profiling: every launch in line on the caller's stream
        # One side stream and one part stream per DEVICE, shared by every plan of the process (a second model -- another
        # config of bench.py, an evaluation copy -- must not add streams: the runtime maps streams onto a handful of hardware
        # queues, and two chains that land on one queue serialise each other).
        cuda = torch.device(device).type == "cuda"
        self.side = _shared("side", device, lambda: SideStream(device)) if (
            cuda and not one and os.environ.get("BMP_WGRAD_STREAM", "1") != "0") else None
        self.split = _shared("part", device, lambda: PartStream(device, int(os.environ.get("BMP_FWD_CHAINS", "2")) - 2)) if (
            cuda and not one and os.environ.get("BMP_FWD_SPLIT", "1") != "0") else None
        self.state: Dict[str, dict] = {}
        self._views()

    def _section(self, prefix, mod, stand, poffs, prep_off, gk_off, tabs_p, grad_lists):
        # ---- forward table: prepared arrays from the flat parameters ----
        prep = _index_eval(mod, "prepared_layouts", stand)
        with torch.no_grad():
            shapes_p = {k: tuple(v.shape) for k, v in mod.prepared_layouts().items()}
        sl = {}
        for name, per_param in prep.items():
            n_el = len(per_param[0])
            terms = []
            for k, a in enumerate(per_param):
                if a.any():
                    terms.append(np.where(a > 0, a - 1 + poffs[k], -1))
            tabs_p.append(_stack_terms(terms, n_el))
            sl[name] = (prep_off, shapes_p[name])
            prep_off += n_el
        self.prep_slices[prefix] = sl
        # ---- backward table: parameter gradients from the kernels' gradient buffers ----
        prim = _index_eval(mod, "primary_layouts", stand)
        spec = mod.gk_spec()
        gsl, gk_idx = {}, {}
        for gname, shp in spec.items():
            n_el = int(np.prod(shp))
            gsl[gname] = (gk_off, tuple(shp))
            gk_idx[gname] = torch.arange(gk_off + 1, gk_off + n_el + 1, dtype=torch.float64).view(shp)
            gk_off += n_el
        self.gk_slices[prefix] = gsl
        pg = mod.primary_grads(gk_idx)
        for name, per_param in prim.items():
            if name not in pg:
                raise KeyError(f"{type(mod).__name__}.primary_grads gives no gradient for layout {name}")
            terms_q = [np.rint(t.reshape(-1).numpy()).astype(np.int64) - 1 for t in pg[name]]     # gk positions
            for k, a in enumerate(per_param):
                nz = np.nonzero(a)[0]
                if len(nz) == 0:
                    continue
                js = a[nz] - 1 + poffs[k]
                for q in terms_q:
                    if len(q) != len(a):
                        raise ValueError(f"gradient of layout {name} has {len(q)} elements, layout has {len(a)}")
                    for j, qq in zip(js.tolist(), q[nz].tolist()):
                        if qq >= 0:
                            grad_lists.setdefault(j, []).append(qq)
        return prep_off, gk_off

    def _views(self) -> None:
        self.P = {pre: {k: self.prep[o:o + int(np.prod(s))].view(s) for k, (o, s) in sl.items()}
                  for pre, sl in self.prep_slices.items()}
        self.G = {pre: {k: self.gk[o:o + int(np.prod(s))].view(s) for k, (o, s) in sl.items()}
                  for pre, sl in self.gk_slices.items()}

    # ---- per step -------------------------------------------------------------------------------
    def prepare(self, flat: torch.Tensor) -> None:
        """One launch: all kernel-layout arrays from the flat parameter buffer.  Also forgets which
        gradient buffers have been written in the previous step."""
        if flat.is_cuda:
            check(_lib.lib().bmp_gather_sum(ptr(self.prep), self.n_prep, ptr(flat), ptr(self.tab_p), self.Kp, 0, stream()),
                  "bmp_gather_sum(prepare)")
        else:       # host form of the same table walk (tests of the tables without a GPU)
            self.prep.copy_(gather_sum_host(flat.detach(), self.tab_p_host))
        if self.state.get("deferred"):               # a planned forward nobody collected (predict, an unplanned co-attention):
            from .functional import flush_deferred   # the readout it held back still runs, its output is not left undefined
            flush_deferred(self.state)
        if self.state.get("side_used"):              # a backward whose gradients nobody collected
            self.state["side"].join()
        if self.state.get("split_open"):
            self.state["split"].join()
        self.state = {}                              # (drops split_keep / the side stream's keep list: both streams are joined)
        # in_line (set by bmp.dp.GraphedTrainStep while it records a step): every launch on the caller's stream.  A replayed
        # HIP graph runs a single in-order chain back to back, while every cross-stream edge of a recording costs tens of
        # microseconds per replay (32-pair step: 1.25 ms recorded in line, 1.52 ms with the side stream and the second chain)
        if self.side is not None and not getattr(self, "in_line", False):
            self.state["side"] = self.side
        if self.split is not None and not getattr(self, "in_line", False):
            self.state["split"] = self.split
        self.gk.zero_()          # a buffer no backward kernel writes this step (an unused readout, ...) must read as zero

    def collect(self, flat_grad: torch.Tensor, overwrite: bool = False) -> None:
        """One launch: flat_grad[j] += the parameter gradients folded out of the kernels' buffers (``overwrite``: = instead
        of +=, zeros where the plan has nothing: for a gradient buffer that holds nothing yet)."""
        if flat_grad.is_cuda:
            if self.state.get("split_open"):
                self.state["split"].join()
                self.state["split_open"] = False
                self.state["split_keep"] = []
            if self.state.get("deferred"):
                from .functional import flush_deferred
                flush_deferred(self.state)
            if self.state.get("deferred_bwd"):
                from .functional import flush_deferred_bwd
                flush_deferred_bwd(self.state)
            if self.state.get("side_used"):          # the side stream's weight gradients land in gk
                self.state["side"].join()
                self.state["side_used"] = False
            check(_lib.lib().bmp_gather_sum(ptr(flat_grad), self.n_flat, ptr(self.gk), ptr(self.tab_g), self.Kg,
                                            0 if overwrite else 1, stream()), "bmp_gather_sum(collect)")
        elif overwrite:
            flat_grad.copy_(gather_sum_host(self.gk, self.tab_g_host))
        else:
            flat_grad.add_(gather_sum_host(self.gk, self.tab_g_host))


def _stack_terms(terms: List[np.ndarray], n_el: int) -> np.ndarray:
    """Per-element compaction of the per-parameter term arrays into [K][n_el] (K = max terms of any element)."""
    if not terms:
        return np.full((1, n_el), -1, dtype=np.int32)
    T = np.stack(terms)                                   # [n_param_terms][n_el], -1 where absent
    order = np.argsort(T < 0, axis=0, kind="stable")      # present entries first, per element
    T = np.take_along_axis(T, order, axis=0)
    K = int((T >= 0).sum(axis=0).max())
    return T[:max(K, 1)].astype(np.int32)


def _concat_tables(tabs: List[np.ndarray]) -> np.ndarray:
    K = max((t.shape[0] for t in tabs), default=1)
    out = [np.concatenate((t, np.full((K - t.shape[0], t.shape[1]), -1, dtype=np.int32))) for t in tabs]
    return np.concatenate(out, axis=1) if out else np.full((1, 0), -1, dtype=np.int32)


def gather_sum_host(src: torch.Tensor, tab: np.ndarray) -> torch.Tensor:
    """Reference semantics of bmp_gather_sum on host tensors."""
    t = torch.from_numpy(tab.astype(np.int64))
    v = src.reshape(-1)[t.clamp(min=0)]
    return torch.where(t >= 0, v, torch.zeros_like(v)).sum(dim=0)
