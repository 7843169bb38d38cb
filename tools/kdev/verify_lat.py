import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ripp_amd as R
R.init(0)
for lg in (12, 13, 14, 15):
    n = 1 << lg
    a, b, r = R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n)
    z = R.product_of_pairings_with_coeffs(a, b, r)
    proof = R.SIPP.prove(a, b, r, z)
    ts = []
    for _ in range(6):
        t = time.perf_counter(); ok = R.SIPP.verify(a, b, r, z, proof); ts.append((time.perf_counter() - t) * 1e3)
    print(n, ok, " ".join("%.1f" % t for t in ts), flush=True)
