from bmp.ggnn import GGNN, MAX_ATOMIC_NUM  # noqa: F401  (models/ggnn_att.py: GGNN + get_atom_array)
