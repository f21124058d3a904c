from bmp.bimpm import BiMPM  # noqa: F401  (train_binary.py:50: from models.coattention.bimpm import BiMPM)
