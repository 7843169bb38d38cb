"""ctypes loader of libripp_hip.so -- the ONLY compute backend of this package.

There is deliberately no CPU fallback: if the HIP library is missing this raises, and every compute entry
point of the library itself returns RIPP_ERR_DEVICE when no gfx950 device is usable.
"""
import ctypes
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "lib", "libripp_hip.so")

RIPP_OK, RIPP_ERR_LENGTH, RIPP_ERR_POW2, RIPP_ERR_DEVICE, RIPP_ERR_ARG = 0, 1, 2, 3, 4


class RippStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in (
        "total_ms", "upload_ms", "scale_ms", "miller_lines_ms", "miller_products_ms", "fold_ms", "normalize_ms",
        "host_ms", "hash_ms", "kernel_miller_lines_ms_sum", "kernel_line_products_ms_sum")] + [
        (n, ctypes.c_uint64) for n in ("kernel_miller_lines_launches", "kernel_line_products_launches", "pairs_lines", "pairs_products")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def lib():
    """Load (once) and return the HIP engine library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  ripp_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.ripp_last_error.restype = ctypes.c_char_p
        for name in ("ripp_ser_gt", "ripp_ser_g1", "ripp_ser_g2", "ripp_ser_fr", "ripp_sipp_job_rounds_left", "ripp_sipp_job_local_len"):
            getattr(L, name).restype = ctypes.c_size_t
        _lib = L
    return _lib


def last_error():
    return lib().ripp_last_error().decode()
