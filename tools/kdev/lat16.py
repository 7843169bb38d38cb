import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import ripp_amd as R
R.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
a, b, r = R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n)
z = R.product_of_pairings_with_coeffs(a, b, r)
for _ in range(3): R.SIPP.prove(a, b, r, z)
t = time.perf_counter()
for _ in range(reps): R.product_of_pairings_with_coeffs(a, b, r)
print("direct ms", (time.perf_counter() - t) / reps * 1e3)
t = time.perf_counter()
for _ in range(reps): out = R.SIPP.prove_with_stats(a, b, r, z)
print("prove ms", (time.perf_counter() - t) / reps * 1e3, out[-1])
t = time.perf_counter()
for _ in range(reps): R.product_of_pairings(a, b)
print("pp ms", (time.perf_counter() - t) / reps * 1e3)
