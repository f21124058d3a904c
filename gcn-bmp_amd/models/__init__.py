"""Import aliases with the reference's module paths (``from models.ggnn import GGNN`` ...), so the
reference's trainer scripts find the MI355X-native operators under the names they already use
(models/__init__.py:7-10 of the reference).  The implementations live in ``bmp``."""
from bmp.ggnn import GGNN                       # noqa: F401
from bmp.relgcn import RelGCN                   # noqa: F401
from bmp.mlp import MLP                         # noqa: F401
