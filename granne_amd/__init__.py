"""granne_amd -- MI355X-native search path for granne (Granne::search on gfx950).

Host-side mirror of the reference's Python interface (py/src/lib.rs: classes Granne and
GranneBuilder, function compute_distance) over the C ABI in include/granne_hip.h.
"""
from ._lib import F32, I8, UNUSED, GranneHipError  # noqa: F401
from .index import Granne, compute_distance, normalize, quantize  # noqa: F401
from .builder import GranneBuilder  # noqa: F401

__all__ = ["Granne", "GranneBuilder", "compute_distance", "normalize", "quantize", "GranneHipError", "F32", "I8", "UNUSED"]
