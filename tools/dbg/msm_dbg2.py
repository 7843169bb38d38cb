import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np, orclib as o, ripp_amd as R
R.init(0)
n = 1
b1 = o.gen_g1(5, n)
for k in (1, 2, 3, 4, 5, 8, 16):
    s = o.fr_array([k] * n)
    got = R.normalize_batch_g1(R.MultiexponentiationInnerProductG1.inner_product(o.blind_g1(b1, 9), s))
    hits = [m for m in range(0, 200) if np.array_equal(got, o.g1_to_affine(o.msm_g1_a(b1, o.fr_array([m]))).reshape(1, 12))]
    print(k, hits, got[0][:2], flush=True)
