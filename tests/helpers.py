"""Fixture decoding shared by CPU and GPU tests."""
import numpy as np
import orclib as o


def g1pt(v): return None if v is None else (int(v[0], 16), int(v[1], 16))
def g2pt(v): return None if v is None else ((int(v[0][0], 16), int(v[0][1], 16)), (int(v[1][0], 16), int(v[1][1], 16)))
def g1arr(vs): return o.g1_array([g1pt(v) for v in vs])
def g2arr(vs): return o.g2_array([g2pt(v) for v in vs])
def frarr(vs): return o.fr_array([int(v, 16) for v in vs])


def gt_from_bytes(hexstr):
    """576-byte serialize_uncompressed image -> (72,) Montgomery limbs."""
    b = bytes.fromhex(hexstr)
    out = np.zeros(72, dtype=np.uint64)
    for i in range(12):
        out[6 * i:6 * i + 6] = o.fp_to_limbs(int.from_bytes(b[48 * i:48 * i + 48], "little"))
    return out
