# scale probe: SIPP prove at growing n, GPU only (no oracle comparison at big sizes), with phase stats
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import ripp_amd as R
R.init(0)
for lg in [int(x) for x in sys.argv[1:]] or [10, 14]:
    n = 1 << lg
    t = time.time(); a = R.synth_g1(1000, n); b = R.synth_g2(2000, n); r = R.synth_fr(0, n); tg = time.time() - t
    t = time.time(); v = R.product_of_pairings_with_coeffs(a, b, r); tv = time.time() - t
    job = R.SippJob(a, b, r)
    t = time.time(); proof, ch, st = job.prove(v); dt = time.time() - t
    print("n=2^%d gen %.2fs direct %.3fs prove %.3fs -> %.0f pairs/s" % (lg, tg, tv, dt, n / dt), flush=True)
    print("   ", {k: round(v, 2) if isinstance(v, float) else v for k, v in st.items()}, flush=True)
    job.close()
