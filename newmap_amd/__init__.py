"""newmap_amd -- MI355X-native engine for newmap's `search` hot path.

Host side mirrors the reference's Python interface for this path (same module and function
names, argument meaning and exceptions); the arithmetic runs in hand-written HIP kernels behind
the C-ABI of include/newmap_amd.h (libnewmap_amd.so).  There is no CPU fallback.
"""
__version__ = "0.1.0"
