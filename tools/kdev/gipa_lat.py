import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import ripp_amd as R, ripp_amd.gipa as G
R.init(0)
def jac1(a): o = np.zeros((len(a), 18), dtype=np.uint64); o[:, :12] = a; o[:, 12:] = R.api._fp_one() if hasattr(R, "api") else 0; return o
import ripp_amd.api as api
one = api._fp_one()
for n in (64, 1024, 1 << 14):
    a = np.zeros((n, 18), dtype=np.uint64); a[:, :12] = R.synth_g1(11, n); a[:, 12:] = one
    b = np.zeros((n, 36), dtype=np.uint64); b[:, :24] = R.synth_g2(22, n); b[:, 24:30] = one
    ka = np.zeros((n, 36), dtype=np.uint64); ka[:, :24] = R.synth_g2(33, n); ka[:, 24:30] = one
    kb = np.zeros((n, 18), dtype=np.uint64); kb[:, :12] = R.synth_g1(44, n); kb[:, 12:] = one
    args = (G.PairingIP, G.AFGHOCommitmentG1, G.AFGHOCommitmentG2, G.IdentityCommitment(G.GT))
    for res in (True, False):
        g = G.GIPA(*args, resident=res)
        g.prove_with_aux((a, b), (ka, kb, [None]))
        t = time.perf_counter(); g.prove_with_aux((a, b), (ka, kb, [None])); dt = time.perf_counter() - t
        print(f"generic GIPA (pairing) n={n} resident={res}: {dt*1e3:.1f} ms")
    t = time.perf_counter(); R.GIPA_TIPP.prove_with_aux(a, b, ka, kb); print(f"   fused ripp_gipa_tipp_prove n={n}: {(time.perf_counter()-t)*1e3:.1f} ms")
