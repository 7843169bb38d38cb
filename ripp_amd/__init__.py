"""ripp_amd -- MI355X-native inner-pairing-product engine (BLS12-381) behind the reference's trait surface.

Layout:  csrc/ (HIP kernels, host driver, C ABI)   lib/libripp_hip.so (built in-tree)   api.py (host-side mirror
of the reference's InnerProduct / DoublyHomomorphicCommitment / SIPP interfaces over the C ABI).
"""
from . import _lib  # noqa: F401
from .api import *  # noqa: F401,F403
