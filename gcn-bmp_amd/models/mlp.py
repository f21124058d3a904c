from bmp.mlp import MLP  # noqa: F401  (models/mlp.py:20-45)
from bmp.link import NTN, DistMult, SymMLP, HolE, BilinearDiag  # noqa: F401  (models/mlp.py:48-193)
