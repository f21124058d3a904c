from bmp.coattention import NieFineCoattention, VQAParallelCoattention  # noqa: F401  (models/coattention/__init__.py)
