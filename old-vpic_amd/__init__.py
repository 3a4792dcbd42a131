"""MI355X-native VPIC inner-loop engine (advance_p, field stencils, accumulator/interpolator glue).

The product is libvpic_hip.so (hand-written HIP for gfx950 behind the C ABI of include/vpic_hip.h);
this package is the thin host-side mirror used by tests, bench.py and the multi-GPU driver.
"""
from . import layout  # noqa: F401
from ._lib import build, lib  # noqa: F401
from .engine import Engine, GridDesc, VpicHipError, make_grid  # noqa: F401
