"""roms_amd -- MI355X-native implementation of the ROMS nonlinear 3-D time step.

The compute path is libroms_hip.so (hand-written HIP kernels for gfx950 behind the C ABI of
include/roms_hip.h).  This package only loads it and mirrors the reference's kernel(ng,tile)
interface for tests and benchmarks; there is no CPU fallback: loading fails loudly when the
library or a GPU is missing.
"""
from . import hiplib  # noqa: F401
