import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, orclib as o, ripp_amd as R
n=1<<int(sys.argv[1]) if len(sys.argv)>1 else 1<<17
a,b,r=o.gen_g1(1,n),o.gen_g2(2,n),o.gen_scalars(3,n); v=o.gt_one()
t=time.time(); d=R.sipp_seed_digest(a,b,r,v); dt=time.time()-t
print("digest: %.3fs -> %.0f MB/s ; ok=%s"%(dt, n*320/dt/1e6, d==o.sipp_seed_digest(a,b,r,v)))

import ctypes
hm=ctypes.c_double(); wm=ctypes.c_double()
from ripp_amd._lib import lib
lib().ripp_statement_hash_times(ctypes.byref(hm), ctypes.byref(wm)); print("hash %.1f ms, wait-for-serialisation %.1f ms"%(hm.value, wm.value))
