"""The same host-side mirror (ripp_amd.api) over the BLS12-377 build of the engine, ripp_amd/lib/libripp_hip_377.so -- the curve of the
reference's own SIPP test and `scaling-ipp` example (sipp/src/lib.rs:229, sipp/examples/scaling-ipp.rs:2,10).

    import ripp_amd.bls12_377 as R377
    R377.init(0); proof = R377.SIPP.prove(a, b, r, value)

Same C ABI, same array layouts (12 x u32 Fp limbs, 8 x u32 Fr limbs, Montgomery form with R = 2^384 / 2^256); group elements and scalars are
BLS12-377's.  That build has the GLV / GLS constants and the field-VM programs of its own tower (tools/gen_params.py, tools/vmgen.py) and runs the same
schedule on the same carry-free 14 x 28-bit throughput kernels (since build round 4: fq_curve2.hpp FQ2_BETA for u^2 = -5, the D-type twist's line placement in
fq_miller.hpp / fq_line_products.hpp); TIPA / aggregate_proofs work on it too, the wire format is ark-ec's generic SWFlags layout (wire.hpp).  No CPU fallback either: a missing library or device fails loudly."""
import importlib.util
import os
import sys

from . import _lib as _base

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libripp_hip_377.so")
_lib377 = None


def _lib():
    global _lib377
    if _lib377 is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950 -DRIPP_BLS12_377)")
        _lib377 = _base.load_library(LIB_PATH)
    return _lib377


_spec = importlib.util.spec_from_file_location("ripp_amd._api377", os.path.join(os.path.dirname(os.path.abspath(__file__)), "api.py"))
_m = importlib.util.module_from_spec(_spec)
_m.__package__ = "ripp_amd"
_spec.loader.exec_module(_m)
_m.lib = _lib
_m.last_error = lambda: _lib().ripp_last_error().decode()
_m.LIB_PATH = LIB_PATH
# Fr::one() in Montgomery form for this scalar field
_x = 0x8508C00000000001
_r = _x**4 - _x**2 + 1
import numpy as _np  # noqa: E402
_one = (1 << 256) % _r
_m.FR_ONE = _np.array([(_one >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=_np.uint64)
_m.R_MOD = _r
sys.modules[__name__] = _m
