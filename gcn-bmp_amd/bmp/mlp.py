"""Link predictor ``MLP`` with the reference's signature (models/mlp.py:20-45).

(B, 2*out_dim) -> 32 -> 16 -> class_num on B rows is < 0.1 % of the path's flops
(SURVEY.md 2.3 K10); it runs as plain torch ops on the device.
"""
from __future__ import annotations

import torch
from torch import nn

from .ggnn import Linear


class MLP(nn.Module):
    def __init__(self, out_dim, hidden_dims=(32, 16), activation=torch.relu, in_dim=None):
        """``in_dim`` replaces Chainer's lazy ``Linear(None, ...)`` shape inference."""
        super().__init__()
        if in_dim is None:
            raise ValueError("MLP needs in_dim (Chainer infers it at the first call; torch cannot)")
        dims = [in_dim] + list(hidden_dims)
        self.layers = nn.ModuleList([Linear(dims[i], dims[i + 1]) for i in range(len(hidden_dims))])
        self.l_out = Linear(dims[-1], out_dim)
        self.activation = activation

    def forward(self, x):
        h = x
        for l in self.layers:                                   # models/mlp.py:42-43
            h = self.activation(torch.nn.functional.linear(h, l.W, l.b))
        return torch.nn.functional.linear(h, self.l_out.W, self.l_out.b)
