from bmp.coattention import NieFineCoattention, VQAParallelCoattention, PoolingFineCoattention  # noqa: F401  (models/coattention/__init__.py)
from bmp.coattention import DeepNieFineCoattention, VeryDeepNieFineCoattention, ExtremeDeepNieFineCoattention, FourierFineCoattention  # noqa: F401
from bmp.coarse import ParallelCoattention, CircularParallelCoattention, AlternatingCoattention, GlobalCoattention, NeuralCoattention  # noqa: F401
from bmp.bimpm import BiMPM  # noqa: F401  (models/coattention/bimpm.py)
