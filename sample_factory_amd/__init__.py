"""MI355X-native APPO hot path behind Sample Factory's plugin surface.

The compute path is libsf_hip.so (hand-written HIP for gfx950, C ABI in include/sf_hip.h).  There is no CPU or
PyTorch fallback for it: importing ``sample_factory_amd.lib`` without the built library raises.
"""
__version__ = "0.1.0"
