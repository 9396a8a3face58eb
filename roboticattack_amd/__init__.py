"""roboticattack_amd — MI355X-native adversarial-patch optimisation engine (UADA / UPA / TMA hot path).

The product path is: Python host (this package) -> ctypes -> libvaa_hip.so (C-ABI, include/vaa.h)
-> hand-written HIP kernels for gfx950. There is NO CPU fallback: importing `roboticattack_amd.ops`
on a machine without the built library raises, and calling an op without a GPU raises.
"""
__version__ = "0.1.0"
