"""randlapack_amd -- MI355X-native sketch-and-factor path behind RandLAPACK's driver/comp API.

The product is librlhip.so (hand-written HIP for gfx950, C ABI in include/rlhip.h) plus the C++ driver
layer in include/RandLAPACK_amd/.  This Python package is plumbing for tests, bench.py and smoke().
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
