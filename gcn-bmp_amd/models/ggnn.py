from bmp.ggnn import GGNN, MAX_ATOMIC_NUM  # noqa: F401  (models/ggnn.py)
