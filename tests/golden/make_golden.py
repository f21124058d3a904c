"""Generate the golden vectors in this directory from the float64 CPU oracle.

The reference itself cannot run here (no chainer / chainer_chemistry / rdkit;
Python-2-only code), and it ships no fixtures of its own, so these vectors are
produced by ``oracle/ref_cpu.py`` (PARITY UNPINNED, see its header).  They pin
the oracle against regressions and give the GPU tests committed
inputs/outputs that do not depend on anything outside the repo.

    python tests/golden/make_golden.py [name filter ...]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gcn-bmp_amd"))

from oracle import ref_cpu as O          # noqa: E402
from bmp import synth                    # noqa: E402


def tiny_mols():
    M = synth.Molecule
    i32 = lambda x: np.asarray(x, dtype=np.int32)
    return [
        M(i32([6, 8]), i32([[0, 1, 1]])),                                               # C=O
        M(i32([6, 6, 7, 6, 6]), i32([[0, 1, 3], [1, 2, 3], [2, 3, 3], [3, 4, 3], [4, 0, 3]])),   # aromatic 5-ring
        M(i32([6, 6, 8, 7, 6, 17, 6]), i32([[0, 1, 0], [1, 2, 1], [1, 3, 0], [3, 4, 0], [4, 5, 0], [4, 6, 2]])),
        M(i32([11]), np.zeros((0, 3), np.int32)),                                        # isolated atom, no bonds
    ]


def grads_of(loss, p):
    names = sorted(p)
    gs = torch.autograd.grad(loss, [p[n] for n in names], allow_unused=True)
    return {f"grad:{n}": (g if g is not None else torch.zeros_like(p[n])).numpy() for n, g in zip(names, gs)}


ONLY = sys.argv[1:]          # optional name filters: write only the files whose name contains one of them


def save(name, **arrs):
    if ONLY and not any(f in name for f in ONLY):
        return
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print(name, {k: v.shape for k, v in arrs.items() if not k.startswith(("param:", "grad:"))})


def main():
    T = torch.from_numpy
    mols = tiny_mols()
    rs = np.random.RandomState(11)

    # ---- GGNN encoder, 4 molecules padded to A=7, d=8, out=8, T=3, tied -------------------
    atoms, adj = synth.concat_mols(mols)
    dr = O._Draw(777, torch.float64, 0.1)
    O.init_ggnn(dr, "", 8, 8, 3)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    g, at = O.ggnn_forward(p, T(atoms), T(adj).double(), 3)
    cg = T(rs.normal(size=g.shape)); ca = T(rs.normal(size=at.shape))
    loss = (g * cg).sum() + 0.1 * (at * ca).sum()
    save("ggnn_small.npz", atoms=atoms, adj=adj, g=g.detach().numpy(), atom_out=at.detach().numpy(),
         cg=cg.numpy(), ca=ca.numpy(), loss=loss.detach().numpy(),
         **{f"param:{k}": v.detach().numpy() for k, v in p.items()}, **grads_of(loss, p))

    # ---- GGNN untied, 2 layers, extra padding (A=10) -------------------------------------
    A = 10
    atoms_p = np.zeros((4, A), np.int32); atoms_p[:, :7] = atoms
    adj_p = np.zeros((4, 4, A, A), np.float32); adj_p[:, :, :7, :7] = adj
    dr = O._Draw(778, torch.float64, 0.1)
    O.init_ggnn(dr, "", 8, 8, 2, weight_tying=False)
    p = {k: v.requires_grad_() for k, v in dr.p.items()}
    g, at = O.ggnn_forward(p, T(atoms_p), T(adj_p).double(), 2, weight_tying=False)
    loss = (g * cg).sum()
    save("ggnn_untied_pad10.npz", atoms=atoms_p, adj=adj_p, g=g.detach().numpy(), atom_out=at.detach().numpy(),
         cg=cg.numpy(), loss=loss.detach().numpy(),
         **{f"param:{k}": v.detach().numpy() for k, v in p.items()}, **grads_of(loss, p))

    # ---- pair predictor: GGNN + Nie co-attention + MLP, 3 pairs ----------------------------
    s1 = [mols[0], mols[2], mols[3]]; s2 = [mols[1], mols[1], mols[2]]
    a1, j1 = synth.concat_mols(s1); a2, j2 = synth.concat_mols(s2)
    label = np.array([[1], [0], [1]], np.int32)
    p = O.make_pair_params(779, hidden_dim=8, out_dim=8, n_layers=2, attn="nie", head=8, dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=2, attn="nie")
    loss = O.sigmoid_cross_entropy(y, T(label))
    save("pair_nie_small.npz", atoms_1=a1, adj_1=j1, atoms_2=a2, adj_2=j2, label=label,
         y=y.detach().numpy(), g1=g1.detach().numpy(), g2=g2.detach().numpy(), loss=loss.detach().numpy(),
         **{f"param:{k}": v.detach().numpy() for k, v in p.items()}, **grads_of(loss, p))

    # ---- pair predictor: RelGCN(3 x 8) + Nie + MLP ---------------------------------------------
    p = O.make_pair_params(780, encoder="relgcn", hidden_dim=8, out_dim=8, n_layers=3, attn="nie", head=8,
                           dtype=torch.float64)
    p = {k: v.requires_grad_() for k, v in p.items()}
    y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), encoder="relgcn", n_layers=3)
    loss = O.sigmoid_cross_entropy(y, T(label))
    save("pair_relgcn_small.npz", atoms_1=a1, adj_1=j1, atoms_2=a2, adj_2=j2, label=label,
         y=y.detach().numpy(), g1=g1.detach().numpy(), g2=g2.detach().numpy(), loss=loss.detach().numpy(),
         **{f"param:{k}": v.detach().numpy() for k, v in p.items()}, **grads_of(loss, p))

    # ---- the other co-attention families on the same three pairs (GGNN 2 x 8 encoder, head 8 / 1) ------------
    for k, attn in enumerate(("deep", "extreme-deep", "fourier", "circ", "pool", "parallel", "alternating", "global", "neural",
                              "bimpm")):
        p = O.make_pair_params(790 + k, hidden_dim=8, out_dim=8, n_layers=2, attn=attn, head=8, dtype=torch.float64)
        p = {kk: v.requires_grad_() for kk, v in p.items()}
        y, g1, g2 = O.pair_forward(p, T(a1), T(j1).double(), T(a2), T(j2).double(), n_layers=2, attn=attn)
        loss = O.sigmoid_cross_entropy(y, T(label))
        save(f"pair_attn_{attn.replace('-', '_')}.npz", atoms_1=a1, adj_1=j1, atoms_2=a2, adj_2=j2, label=label,
             y=y.detach().numpy(), g1=g1.detach().numpy(), g2=g2.detach().numpy(), loss=loss.detach().numpy(),
             **{f"param:{kk}": v.detach().numpy() for kk, v in p.items()}, **grads_of(loss, p))


if __name__ == "__main__":
    main()
