"""dynosam_amd — MI355X-native Levenberg-Marquardt factor-graph solver behind DynoSAM's
GTSAM-facing backend seam.  Compute lives in csrc/libdynogfx.so (hand-written gfx950 HIP,
C-ABI in include/dynogfx.h); this package is the host-side mirror of the reference interface."""
from . import graph, symbols, synth  # noqa: F401
from .graph import FlatGraph, FactorBlock  # noqa: F401


def __getattr__(name):
    if name in ("LevenbergMarquardtOptimizer", "LevenbergMarquardtParams", "Context"):
        from . import optimizer
        return getattr(optimizer, name)
    raise AttributeError(name)
