"""MI355X-native bidirectional path tracing backend: harness-side Python plumbing.

The product is the HIP library built from csrc/ (libubdpt.so, C ABI in include/ubdpt.h) plus the
C++ host layer that mirrors the reference's URenderer/UEngine interface. This package only loads
that library through ctypes for tests, bench.py and the smoke run, and generates the synthetic
scenes. It contains no rendering code and no CPU fallback.
"""
from . import capi, multigpu, scenes  # noqa: F401

__all__ = ["capi", "multigpu", "scenes"]
