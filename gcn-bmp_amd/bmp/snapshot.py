"""Parameter exchange with the reference's link-tree naming.

A Chainer snapshot (``serializers.save_npz``, train_binary.py:664) stores parameters under
slash-separated link paths (``graph_conv/message_layers/0/W``).  The modules in this package
use the same attribute names, so ``a/b/0/W`` maps to ``a.b[0].W``.
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np
import torch
from torch import nn


def _resolve(module: nn.Module, path: str):
    obj = module
    for part in path.split("/"):
        obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
    return obj


def load_param_dict(module: nn.Module, params: Mapping[str, "torch.Tensor | np.ndarray"], prefix: str = "",
                    strict: bool = True) -> None:
    """Copy ``params[prefix + name]`` into the module's parameters (cast to float32)."""
    seen = set()
    for key, val in params.items():
        if not key.startswith(prefix):
            continue
        name = key[len(prefix):]
        try:
            p = _resolve(module, name)
        except (AttributeError, IndexError, TypeError):
            if strict:
                raise KeyError(f"no parameter {name!r} in {type(module).__name__}")
            continue
        v = torch.as_tensor(np.asarray(val.detach().cpu()) if isinstance(val, torch.Tensor) else val)
        if tuple(v.shape) != tuple(p.shape):
            raise ValueError(f"{name}: shape {tuple(v.shape)} != {tuple(p.shape)}")
        with torch.no_grad():
            p.copy_(v.to(p.dtype))
        seen.add(name.replace("/", "."))
    if strict:
        missing = [n for n, _ in module.named_parameters() if n not in seen]
        if missing:
            raise KeyError(f"parameters not provided: {missing}")


def param_dict(module: nn.Module, prefix: str = "") -> Dict[str, torch.Tensor]:
    """Inverse of load_param_dict: {prefix + 'a/b/0/W': tensor}."""
    return {prefix + n.replace(".", "/"): p.detach() for n, p in module.named_parameters()}


def grad_dict(module: nn.Module, prefix: str = "") -> Dict[str, torch.Tensor]:
    return {prefix + n.replace(".", "/"): (p.grad.detach() if p.grad is not None else torch.zeros_like(p))
            for n, p in module.named_parameters()}
