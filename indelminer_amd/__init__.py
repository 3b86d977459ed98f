"""indelminer_amd -- MI355X (gfx950) split-read hot path of indelMINER.

The product is libindelminer_amd.so (HIP kernels behind the C ABI declared in
include/indelminer_amd.h) plus the C host driver; this package only holds the
build recipe and a ctypes mirror of the ABI for tests and bench.py.
"""
from . import build  # noqa: F401
from . import capi  # noqa: F401
