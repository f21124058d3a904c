from bmp.relgcn import GGNNModular as GGNN  # noqa: F401  (models/models/ggnn.py)
