import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, orclib as o, ripp_amd as R
R.init(0)
n = 1
a, b = o.gen_g1(123, n), o.gen_g2(456, n)
for v in (2, 16):
    r = o.fr_array([v] * n)
    exp = o.product_of_pairings_with_coeffs(a, b, r)
    got = R.product_of_pairings_with_coeffs(a, b, r)
    print(hex(v), np.array_equal(exp, got), flush=True)
