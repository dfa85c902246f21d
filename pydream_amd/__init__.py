"""pydream_amd -- MI355X-native MT-DREAM(ZS) engine behind PyDREAM's Python API.

Host code is Python (as in the reference); the per-chain hot path runs as hand-written HIP
kernels for gfx950 reached through the C ABI of include/dreamzs.h (ctypes, no PyTorch).
"""
__version__ = "0.1.0"
