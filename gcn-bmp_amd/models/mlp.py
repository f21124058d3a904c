from bmp.mlp import MLP  # noqa: F401  (models/mlp.py)
