"""MI355X-native paired-molecule message passing + co-attention (GCN-BMP hot path).

Host side (Python on PyTorch-ROCm tensors) above the C-ABI library built from
``gcn-bmp_amd/csrc``.  Importing this package does not touch the GPU; the HIP
library is loaded on first use by ``bmp._lib`` and its absence is a hard error.
"""
__version__ = "0.1.0"
