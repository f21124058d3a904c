from bmp.coattention import NieFineCoattention, VQAParallelCoattention, PoolingFineCoattention  # noqa: F401  (models/coattention/__init__.py)
from bmp.coarse import ParallelCoattention, AlternatingCoattention, GlobalCoattention, NeuralCoattention  # noqa: F401
