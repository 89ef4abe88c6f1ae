"""mashmap_amd -- MI355X-native sketch + L1/L2 hot path of MashMap behind a C ABI (include/mashmap_hip.h).

Python here is plumbing for tests and bench.py only (ctypes loader in capi.py).
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
