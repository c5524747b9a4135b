"""lwm_amd -- MI355X-native RingAttention / VQGAN hot path of LWM.

Hand-written HIP (gfx950) kernels behind a C ABI (include/lwm_hip.h), called
from PyTorch-ROCm tensors.  See DESIGN.md.
"""
__version__ = "0.1.0"
