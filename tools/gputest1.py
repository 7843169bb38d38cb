import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import orclib as o
import ripp_amd as R
R.init(0)
def chk(name, got, exp):
    ok = np.array_equal(got, exp); print(("OK   " if ok else "FAIL ") + name, flush=True); return ok
allok = True
# synth vs oracle
n = 300
a = o.gen_g1(1000, n); b = o.gen_g2(2000, n); r = o.gen_scalars(0, n)
allok &= chk("synth_g1", R.synth_g1(1000, n), a)
allok &= chk("synth_g2", R.synth_g2(2000, n), b)
allok &= chk("synth_fr", R.synth_fr(0, n), r)
# normalize
aj = o.blind_g1(a, 7); bj = o.blind_g2(b, 8); aj[3] = 0; aj[3, 0] = 5  # a z=0 point
allok &= chk("normalize_g1", R.normalize_batch_g1(aj), o.normalize_g1(aj))
allok &= chk("normalize_g2", R.normalize_batch_g2(bj), o.normalize_g2(bj))
# scale
allok &= chk("scale_g1", R.scale_g1_affine(a, r), o.scale_g1_a(a, r))
# folds
s = r[5]
allok &= chk("fold_g1_a", R.fold_g1_affine(a[150:], a[:150], s), o.fold_g1_a(a[150:], a[:150], s))
allok &= chk("fold_g2_a", R.fold_g2_affine(b[150:], b[:150], s), o.fold_g2_a(b[150:], b[:150], s))
allok &= chk("fold_g1_j", R.normalize_batch_g1(R.fold_g1(aj[150:], aj[:150], s)), o.normalize_g1(o.fold_g1_j(aj[150:], aj[:150], s)))
allok &= chk("fold_g2_j", R.normalize_batch_g2(R.fold_g2(bj[150:], bj[:150], s)), o.normalize_g2(o.fold_g2_j(bj[150:], bj[:150], s)))
# pairing products
for m in (1, 2, 5, 64, 300):
    t = time.time(); got = R.product_of_pairings(a[:m], b[:m]); dt = time.time() - t
    allok &= chk("product_of_pairings n=%d (%.3fs)" % (m, dt), got, o.pairing_product_a(a[:m], b[:m]))
rc, exp = o.pairing_product_j(aj, bj)
allok &= chk("PairingInnerProduct n=300 (jacobian, one inf)", R.PairingInnerProduct.inner_product(aj, bj), exp)
allok &= chk("product_with_coeffs", R.product_of_pairings_with_coeffs(a[:64], b[:64], r[:64]), o.product_of_pairings_with_coeffs(a[:64], b[:64], r[:64]))
allok &= chk("empty product", R.product_of_pairings(a[:0], b[:0]), o.pairing_product_a(a[:0], b[:0]))
try:
    R.PairingInnerProduct.inner_product(aj[:5], bj[:4]); print("FAIL length error"); allok = False
except R.InnerProductError as e:
    print("OK   length error:", e)
# SIPP
for m in (2, 16, 256):
    A, B, RR = a[:m], b[:m], r[:m]
    v = o.product_of_pairings_with_coeffs(A, B, RR)
    t = time.time(); proof, ch, st = R.SIPP.prove_with_stats(A, B, RR, v); dt = time.time() - t
    rc, eproof, ech = o.sipp_prove(A, B, RR, v)
    allok &= chk("SIPP prove n=%d proof (%.3fs)" % (m, dt), proof, eproof)
    allok &= chk("SIPP prove n=%d challenges" % m, ch, ech)
print("ALL OK" if allok else "SOME FAILED")
print(st)
