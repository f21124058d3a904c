"""``GraphConvPredictorForPair`` -- the pair glue the reference re-declares in every script
(no co-attention: train_ddi_modify.py:46-82; with co-attention: train_binary.py:59-141 =
eval_coattention.py:41-126)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .ggnn import GGNN, PackedAtoms, as_packed
from .mlp import MLP, sigmoid_cross_entropy      # noqa: F401  (re-exported: the pair loss)
from .packed import PackedMolBatch


class GraphConvPredictorForPair(nn.Module):
    def __init__(self, graph_conv, attn=None, mlp=None, symmetric=None):
        super().__init__()
        # train_ddi_modify.py:47 declares (graph_conv, mlp=None) and calls GraphConvPredictorForPair(ggnn, mlp) (:150);
        # train_binary.py:60 declares (graph_conv, attn=None, mlp=None, symmetric=None).  Both positional forms work:
        if mlp is None and getattr(attn, "is_link_predictor", False):
            attn, mlp = None, attn
        self.graph_conv = graph_conv
        self.attn = attn
        self.mlp = mlp
        self.symmetric = symmetric
        # The reference builds its link predictors with Chainer's lazy input width (MLP(out_dim, hidden_dims),
        # train_ddi_modify.py:136; SymMLP / HolE, train_binary.py:170,178).  The width is known here -- what the co-attention
        # (or, without one, the encoder) hands over per molecule -- so the parameters exist before an optimizer flattens them.
        fp = getattr(attn if attn is not None else graph_conv, "out_dim", None)
        if attn is None and getattr(graph_conv, "concat_hidden", False):
            fp = fp * graph_conv.n_layers                                   # models/ggnn.py:646-647
        if fp is not None and callable(getattr(mlp, "materialize_input", None)):
            mlp.materialize_input(int(fp))

    def _encode(self, atoms_1, adjs_1, atoms_2, adjs_2):
        """Siamese encoder (train_binary.py:91-94).  A two-sided PackedMolBatch in the first slot
        encodes both sides in ONE pass (same weights, so it is the same computation)."""
        from .enclayout import EncBatch, EncRowsFn
        if isinstance(atoms_1, EncBatch):
            # the batch in the encoder layout (bmp/enclayout.py: real atoms + one pad row per tile, tile heights balanced over
            # the CUs; optionally every distinct molecule once): the encoder runs there, readout and co-attention on the
            # per-instance rows copied from it
            eb = atoms_1
            h, h0 = self.graph_conv.encode_rows(eb.pb_enc)
            rows = EncRowsFn.apply(h, eb)
            rows0 = None if h0 is None else EncRowsFn.apply(h0, eb)
            at = PackedAtoms(rows, eb.pb, None)
            self.graph_conv.atoms = at
            g = self.graph_conv.readout_rows(rows, rows0, eb.pb)
            B = eb.pb.side_mols[1]
            return g[:B], g[B:], at, at, (0, B)
        if isinstance(atoms_1, PackedMolBatch) and len(atoms_1.side_mols) == 3 and atoms_2 is None:
            pb = atoms_1
            g = self.graph_conv(pb)
            at = self.graph_conv.get_atom_array()
            B = pb.side_mols[1]
            return g[:B], g[B:], at, at, (0, B)
        g1 = self.graph_conv(atoms_1, adjs_1)
        at1 = self.graph_conv.get_atom_array()
        g2 = self.graph_conv(atoms_2, adjs_2)
        at2 = self.graph_conv.get_atom_array()
        return g1, g2, at1, at2, (0, 0)

    def forward(self, atoms_1, adjs_1=None, atoms_2=None, adjs_2=None):
        return self._forward(atoms_1, adjs_1, atoms_2, adjs_2, None)

    def forward_loss(self, *inputs, t):
        """``Classifier(predictor, lossfun=F.sigmoid_cross_entropy)(*inputs, t)`` (train_ddi_modify.py:284-286;
        train_binary.py:520-524): the mean sigmoid cross entropy of the pair logits against the labels ``t`` (-1: not counted).
        With an MLP link predictor on the device the link predictor, the loss and their backward are one launch each way
        (bmp.mlp.MLPLossFn); the logits of the call are left in ``self.y``."""
        inputs = tuple(inputs) + (None,) * (4 - len(inputs))
        return self._forward(*inputs, t)

    def _forward(self, atoms_1, adjs_1, atoms_2, adjs_2, t):
        # A co-attention of the fine family replaces the encoder's molecule vectors without reading them (:96 with
        # nie_coattention.py:335-370): the readout is computed all the same, but a planned encoder may take it off the chain
        put = object.__setattr__         # (plain attributes: nn.Module.__setattr__ costs 5 us a time on a 1 ms step)
        put(self.graph_conv, "_readout_off_chain", bool(getattr(self.attn, "ignores_graph_vectors", False)))
        try:
            g1, g2, at1, at2, mol0 = self._encode(atoms_1, adjs_1, atoms_2, adjs_2)
        finally:
            put(self.graph_conv, "_readout_off_chain", False)
        if self.attn is not None:
            g1, g2 = self.attn(at1, g1, at2, g2, mol0=mol0)                  # train_binary.py:96
        fast = getattr(self.graph_conv, "_fast", None)
        if fast is not None:             # a co-attention off the planned path never flushed the readout the encoder held back
            from .functional import flush_deferred
            flush_deferred(fast[2])
        put(self, "g1", g1); put(self, "g2", g2)
        if t is None:
            return self.mlp(g1, g2)      # MLP on [g1 | g2] :98-101 (no concatenation copy); NTN / HolE / ... :102-116
        if callable(getattr(self.mlp, "forward_loss", None)):
            loss, y = self.mlp.forward_loss(g1, g2, t)
            put(self, "y", y)
            return loss
        put(self, "y", self.mlp(g1, g2))
        return self.loss(self.y, t)

    def predict(self, atoms_1, adjs_1=None, atoms_2=None, adjs_2=None):
        """train_binary.py:120-127 (sigmoid under no-backprop)."""
        with torch.no_grad():
            if self.symmetric is None:
                return torch.sigmoid(self.forward(atoms_1, adjs_1, atoms_2, adjs_2))
            if isinstance(atoms_1, PackedMolBatch):
                raise NotImplementedError("symmetric predict needs the dense four-array form")
            t1 = torch.sigmoid(self.forward(atoms_1, adjs_1, atoms_2, adjs_2))
            t2 = torch.sigmoid(self.forward(atoms_2, adjs_2, atoms_1, adjs_1))
            return torch.maximum(t1, t2) if self.symmetric == 'or' else torch.minimum(t1, t2)

    def predict_eval(self, atoms_1, adjs_1=None, atoms_2=None, adjs_2=None):
        """The ``predict`` of eval_coattention.py:103-124 (the evaluation script re-declares the class with this form):
        logits -- no sigmoid -- and the two molecule vectors handed to the link predictor, under no-backprop."""
        with torch.no_grad():
            h = self.forward(atoms_1, adjs_1, atoms_2, adjs_2)
            return h, (self.g1, self.g2)

    @staticmethod
    def loss(y, t):
        return sigmoid_cross_entropy(y, t)


def build_link_predictor(sim_method: str, fp_out_dim: int, class_num: int, net_hidden_dims=(32, 16)):
    """The ``lp`` selection of set_up_predictor, train_binary.py:165-187."""
    if sim_method == "mlp":
        return MLP(class_num, net_hidden_dims, in_dim=2 * fp_out_dim)
    from .link import NTN, DistMult, HolE, SymMLP
    if sim_method == "ntn":
        return NTN(left_dim=fp_out_dim, right_dim=fp_out_dim, out_dim=class_num, ntn_out_dim=8, hidden_dims=net_hidden_dims)
    if sim_method == "symmlp":
        return SymMLP(out_dim=class_num, hidden_dims=net_hidden_dims, fp_dim=fp_out_dim)
    if sim_method == "hole":
        return HolE(out_dim=class_num, hidden_dims=net_hidden_dims, fp_dim=fp_out_dim)
    if sim_method == "dist-mult":
        return DistMult(left_dim=fp_out_dim, right_dim=fp_out_dim, out_dim=class_num, dm_out_dim=8, hidden_dims=net_hidden_dims)
    raise ValueError('[ERROR] Invalid link prediction model: {}'.format(sim_method))     # train_binary.py:188


def build_pair_predictor(hidden_dim=128, out_dim=128, n_layers=4, weight_tying=True, attn: Optional[str] = "nie",
                         head=8, class_num=1, encoder="ggnn", mlp_hidden=(32, 16), sim_method="mlp"):
    """set_up_predictor counterpart (train_binary.py:144-277) for the configs of BASELINE.json."""
    if encoder == "ggnn":
        enc = GGNN(out_dim=out_dim, hidden_dim=hidden_dim, n_layers=n_layers, weight_tying=weight_tying)
    elif encoder == "relgcn":
        from .relgcn import RelGCN
        enc = RelGCN(out_channels=out_dim, ch_list=[hidden_dim] * (n_layers + 1), scale_adj=True)
    else:
        raise ValueError('[ERROR] Invalid graph embedding encoder.')
    a = None
    if attn in ("nie", "vqa"):
        from .coattention import NieFineCoattention
        a = NieFineCoattention(hidden_dim=hidden_dim, out_dim=out_dim, head=head, activation="tanh")
    elif attn == "bimpm":                                                        # train_binary.py:253-256: head = fp_out_dim
        from .bimpm import BiMPM
        a = BiMPM(hidden_dim=hidden_dim, out_dim=out_dim, head=out_dim, with_max_pool=True, with_att_mean=True, with_att_max=True)
    elif attn in ("deep", "very-deep", "extreme-deep"):                         # train_binary.py:229-247
        from . import coattention as C
        cls = {"deep": C.DeepNieFineCoattention, "very-deep": C.VeryDeepNieFineCoattention,
               "extreme-deep": C.ExtremeDeepNieFineCoattention}[attn]
        a = cls(hidden_dim=hidden_dim, out_dim=out_dim, head=head, activation="tanh")
    elif attn == "fourier":                                                      # train_binary.py:249-252
        from .coattention import FourierFineCoattention
        a = FourierFineCoattention(hidden_dim=hidden_dim, out_dim=out_dim, head=head, activation="tanh")
    elif attn == "pool":
        from .coattention import PoolingFineCoattention
        a = PoolingFineCoattention(hidden_dim=hidden_dim, out_dim=out_dim)       # train_binary.py:210-212
    elif attn == "parallel":
        from .coarse import ParallelCoattention
        a = ParallelCoattention(hidden_dim=hidden_dim, out_dim=out_dim, head=1, activation="tanh")   # train_binary.py:200-204
    elif attn == "circ":                                                         # train_binary.py:206-209
        from .coarse import CircularParallelCoattention
        a = CircularParallelCoattention(hidden_dim=hidden_dim, out_dim=out_dim, activation="tanh")
    elif attn == "alternating":
        from .coarse import AlternatingCoattention
        a = AlternatingCoattention(hidden_dim=hidden_dim, out_dim=out_dim, head=head, weight_tying=True)
    elif attn == "global":
        from .coarse import GlobalCoattention
        a = GlobalCoattention(hidden_dim=hidden_dim, out_dim=out_dim)
    elif attn == "neural":
        from .coarse import NeuralCoattention
        a = NeuralCoattention(hidden_dim=hidden_dim, out_dim=out_dim, activation="tanh")
    elif attn is not None:
        raise ValueError('[ERROR] Invalid Co-Attention Method.')
    fp_dim = getattr(a, "out_dim", out_dim) if a is not None else out_dim        # BiMPM hands over 3 * head columns
    return GraphConvPredictorForPair(enc, a, build_link_predictor(sim_method, fp_dim, class_num, mlp_hidden))
