import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, orclib as o, ripp_amd as R
R.init(0)
for n in (1, 2, 3, 5, 16, 64, 1000):
    for vals in (None, [1] * n, [2] * n, [5]*n, [16]*n, [17]*n):
        s = o.gen_scalars(21, n) if vals is None else o.fr_array(vals)
        b1, b2 = o.gen_g1(5, n), o.gen_g2(6, n)
        ok1 = np.array_equal(R.normalize_batch_g1(R.MultiexponentiationInnerProductG1.inner_product(o.blind_g1(b1, 9), s)), o.g1_to_affine(o.msm_g1_a(b1, s)).reshape(1, 12))
        ok2 = np.array_equal(R.normalize_batch_g2(R.MultiexponentiationInnerProductG2.inner_product(o.blind_g2(b2, 9), s)), o.g2_to_affine(o.msm_g2_a(b2, s)).reshape(1, 24))
        print(n, 'rand' if vals is None else vals[0], ok1, ok2, flush=True)
