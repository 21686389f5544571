"""spandsp_amd -- MI355X-native batched tone-detect / modem-demod / echo-cancel engine.

The product is ``libspangpu.so`` (hand-written HIP for gfx950 behind the C ABI in
``include/spangpu.h`` and the spandsp-named C shim in ``include/spangpu_spandsp.h``).
This Python package is only a thin ctypes harness over that ABI for tests and
``bench.py``; there is no Python or CPU implementation of the hot path here, and
loading fails loudly if the library has not been built.
"""
from .engine import (  # noqa: F401
    LIB_PATH,
    SpanGpuError,
    ToneBank,
    device_count,
    goertzel_fac,
    lib,
)
