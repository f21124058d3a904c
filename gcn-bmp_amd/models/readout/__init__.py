from bmp.relgcn import GGNNReadout  # noqa: F401  (models/readout/__init__.py)
