import sys, time, os
sys.path.insert(0, 'tests')
import orclib as o
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 12
a, b, r = o.gen_g1(1000, n), o.gen_g2(2000, n), o.gen_scalars(0, n)
for th in [int(x) for x in sys.argv[2:]] or [1, 16, 64, 128, 256]:
    o.lib().orc_set_num_threads(th)
    t = time.perf_counter(); z = o.pairing_product_a(a, b); t1 = time.perf_counter() - t
    t = time.perf_counter(); f = o.fold_g2_a(b[n//2:], b[:n//2], r[0]); t2 = time.perf_counter() - t
    t = time.perf_counter(); s = o.scale_g1_a(a, r); t3 = time.perf_counter() - t
    print("threads %3d: pairing_product %.3fs (%.1f us/pair/thr)  fold_g2(n/2) %.3fs  scale_g1 %.3fs" % (th, t1, t1 / n * th * 1e6, t2, t3), flush=True)
