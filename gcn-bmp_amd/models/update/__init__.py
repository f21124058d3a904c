from bmp.relgcn import GGNNUpdate, RelGCNUpdate  # noqa: F401  (models/update/__init__.py)
