"""Host-side (CPU) cost of the serial glue between kernels: miller_combine (63 Fp12 squarings + 68 products), final exponentiation, GT pow."""
import sys, time, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from ripp_amd._lib import lib
L = lib()
p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
rng = np.random.default_rng(1)
import orclib as o
f = o.pairing_product_a(o.gen_g1(3, 2), o.gen_g2(5, 2))       # some GT element
rows = np.stack([f] * 68); out = np.zeros(72, dtype=np.uint64)
def best(fn, reps=20):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3
print("miller_combine   %.3f ms" % best(lambda: L.ripp_miller_combine(p(rows), p(out))))
print("final_exp        %.3f ms" % best(lambda: L.ripp_final_exp(p(f), p(out))))
k = o.gen_scalars(1, 1)[0]
print("gt_pow (255 bit) %.3f ms" % best(lambda: L.ripp_gt_pow(p(f), p(k), p(out))))
print("oracle final_exp %.3f ms" % best(lambda: o.final_exp(f)))
