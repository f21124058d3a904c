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


# ---- on-disk formats ---------------------------------------------------------------------
# chainer.serializers.save_npz writes one array per parameter under its slash-separated link path.  A *trainer*
# snapshot (extensions.snapshot(), train_binary.py:664) holds the model under 'updater/model:main/' -- the
# Classifier, whose predictor is the GraphConvPredictorForPair -- and the Adam state of every parameter under
# 'updater/optimizer:main/<path>/{m,v}' plus 'updater/optimizer:main/t'; eval_coattention.py:436-438 loads such a file
# straight into the Classifier.  Both spellings are accepted here.
TRAINER_PREFIX = "updater/model:main/predictor/"
CLASSIFIER_PREFIX = "predictor/"


def _detect_prefix(keys) -> str:
    for pre in (TRAINER_PREFIX, CLASSIFIER_PREFIX, ""):
        if any(k.startswith(pre + "graph_conv/") for k in keys):
            return pre
    raise KeyError("no 'graph_conv/...' parameters found in the snapshot")


def load_chainer_snapshot(path: str, predictor: nn.Module, strict: bool = True) -> str:
    """Load a Chainer npz snapshot (model-only, Classifier or whole trainer) into a GraphConvPredictorForPair.
    Only arrays are read (``numpy.load`` without pickle).  Returns the key prefix that was found."""
    with np.load(path, allow_pickle=False) as z:
        keys = list(z.files)
        pre = _detect_prefix(keys)
        params = {k[len(pre):]: z[k] for k in keys if k.startswith(pre) and not k.startswith("updater/optimizer")}
    load_param_dict(predictor, params, strict=strict)
    return pre


def save_chainer_snapshot(path: str, predictor: nn.Module, prefix: str = TRAINER_PREFIX, adam=None) -> None:
    """Write the predictor's parameters in Chainer's npz key layout (the inverse of load_chainer_snapshot); with
    ``adam`` (a bmp.dp.FlatAdam over the same module) also its step count and first/second moments the way
    chainer.optimizers.Adam serialises them, so training can resume in either framework."""
    out = {prefix + k: v.cpu().numpy() for k, v in param_dict(predictor).items()}
    if adam is not None:
        opt = prefix.replace("model:main/predictor/", "optimizer:main/") if "model:main" in prefix else "optimizer/"
        out[opt + "t"] = np.asarray(adam.t, dtype=np.int32)
        off = 0
        for name, shp in zip(adam.names, adam.shapes):
            n = int(np.prod(shp))
            key = opt + "predictor/" + name.replace(".", "/")
            out[key + "/m"] = adam.m[off:off + n].reshape(shp).cpu().numpy()
            out[key + "/v"] = adam.v[off:off + n].reshape(shp).cpu().numpy()
            off += n
    np.savez(path, **out)


def save_tuple_dataset(path: str, arrays) -> None:
    """NumpyTupleDataset.save (parsers.py:91-104): positional arrays arr_0, arr_1, ... in one npz."""
    np.savez(path, *[np.asarray(a) for a in arrays])


def load_tuple_dataset(path: str):
    """NumpyTupleDataset.load (parsers.py:106-120): the arrays arr_0.. in order, or None when the file is missing."""
    import os
    if not os.path.exists(path):
        return None
    with np.load(path, allow_pickle=False) as z:
        out, i = [], 0
        while f"arr_{i}" in z.files:
            out.append(z[f"arr_{i}"]); i += 1
    return tuple(out)
