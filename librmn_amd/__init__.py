"""librmn_amd -- MI355X-native drop-in for librmn's EZ grid interpolation + field packers hot path.

The product is the C-ABI shared library ``librmn_ez_hip.so`` (C host front-end + hand-written HIP
kernels for gfx950, see csrc/).  This package is the Python host-side mirror of the reference's
interface for that path: same function names, argument meaning and return codes as librmn's
``c_ez*`` / ``c_gd*`` / ``compact_*`` / ``armn_compress`` (include/*.h cite the reference lines).

There is no CPU fallback: importing works anywhere, but every compute call needs the built
library and a HIP device and raises/returns an error otherwise.
"""
from .lib import load_library, library_path, build_library   # noqa: F401
from . import ezscint, packers, interpv                        # noqa: F401

__all__ = ["load_library", "library_path", "build_library", "ezscint", "packers", "interpv"]
