import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import ripp_amd as R
from ripp_amd._lib import lib
R.init(0)
n = 1 << 20
a, b, r = R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n)
z = R.product_of_pairings_with_coeffs(a, b, r)
for it in range(3):
    t = time.perf_counter(); out = R.SIPP.prove_with_stats(a, b, r, z); dt = time.perf_counter() - t
    h, w = ctypes.c_double(), ctypes.c_double(); lib().ripp_statement_hash_times(ctypes.byref(h), ctypes.byref(w))
    print(f"prove {dt*1e3:.1f} ms; digest: hash {h.value:.1f} ms, wait-for-serialisation {w.value:.1f} ms; stats {out[-1]}")
t = time.perf_counter(); d = R.sipp_seed_digest(a, b, r, z); print("standalone digest ms", (time.perf_counter() - t) * 1e3)
h, w = ctypes.c_double(), ctypes.c_double(); lib().ripp_statement_hash_times(ctypes.byref(h), ctypes.byref(w)); print("standalone: hash", h.value, "wait", w.value)
print(open("/proc/cpuinfo").read().count("processor"), "cpus;", [l for l in open("/proc/cpuinfo") if "model name" in l][0].strip())
