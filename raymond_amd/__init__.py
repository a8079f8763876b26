"""raymond_amd — MI355X-native drop-in for the per-pixel radiance loop of Nyrox/raymond.

The product is the C-ABI library built from raymond_amd/csrc (include/raymond_hip.h);
this package is the thin host-side mirror of the reference's scene/settings API used
by the tests, bench.py and the examples.  Importing the package does not load the
HIP library; calling into it does, and fails loudly when it is missing.
"""
from . import abi  # noqa: F401

__all__ = ["abi", "lib", "scene", "scenes", "render"]
