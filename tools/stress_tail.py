"""Randomized stress of the pipelined tail (python tools/stress_tail.py <seed> <seconds> <min log n> <max log n>): many proofs of random small statements against the CPU oracle, interleaved sizes, one process."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orclib as o, ripp_amd as R
R.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0; t0 = time.time(); cnt = 0
while time.time() - t0 < (float(sys.argv[2]) if len(sys.argv) > 2 else 60):
    lg = int(rng.integers(int(sys.argv[3]) if len(sys.argv) > 3 else 1, int(sys.argv[4]) if len(sys.argv) > 4 else 11)); n = 1 << lg
    sa, sb, sr = (int(x) for x in rng.integers(1, 1 << 30, 3))
    a, b, r = o.gen_g1(sa, n), o.gen_g2(sb, n), o.gen_scalars(sr, n)
    if rng.random() < 0.2: a[int(rng.integers(0, n))] = 0          # an identity somewhere
    if rng.random() < 0.2: b[int(rng.integers(0, n))] = 0
    v = o.product_of_pairings_with_coeffs(a, b, r)
    rc, ep, _ = o.sipp_prove(a, b, r, v)
    p = R.SIPP.prove(a, b, r, v)
    ok = rc == 0 and np.array_equal(p, ep) and R.SIPP.verify(a, b, r, v, p)
    cnt += 1
    if not ok: bad += 1; print("MISMATCH", n, sa, sb, sr, flush=True)
print("proofs", cnt, "mismatches", bad)
