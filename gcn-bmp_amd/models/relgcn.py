from bmp.relgcn import RelGCN, rescale_adj  # noqa: F401  (models/relgcn.py)
