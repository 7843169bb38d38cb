"""MSM wall time vs n on one MI355X (device-resident timing is not exposed; this is the C-ABI call incl. H2D of affine bases)."""
import sys, time, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, ripp_amd as R
from ripp_amd._lib import lib
R.init(0)
N = 1 << max([int(x) for x in sys.argv[1:]] or [16])
a, b, s = R.synth_g1(1000, N), R.synth_g2(2000, N), R.synth_fr(2, N)
p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
for lg in [int(x) for x in (sys.argv[1:] or range(0, 17))]:
    n = 1 << lg
    o1 = np.zeros(18, dtype=np.uint64); o2 = np.zeros(36, dtype=np.uint64)
    t1 = []; t2 = []
    for _ in range(4):
        t = time.perf_counter(); lib().ripp_msm_g1_a(p(a), p(s), ctypes.c_size_t(n), p(o1)); t1.append(time.perf_counter() - t)
        t = time.perf_counter(); lib().ripp_msm_g2_a(p(b), p(s), ctypes.c_size_t(n), p(o2)); t2.append(time.perf_counter() - t)
    print(f"n=2^{lg:2d}  G1 {min(t1)*1e3:8.2f} ms   G2 {min(t2)*1e3:8.2f} ms", flush=True)
