"""mesheditor_amd -- MI355X-native modal-audio hot path (FEM assembly + generalised eigensolve + resonator bank).

The compute lives in libmodalhip.so (hand-written HIP for gfx950 behind the C ABI of include/modalhip.h).
This package is the thin Python binding used by the tests and bench.py; there is no CPU fallback.
"""
from . import meshes  # noqa: F401

__all__ = ["meshes"]
