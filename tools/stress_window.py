"""Randomized stress of the hash-window machinery (python tools/stress_window.py <seed> <seconds> <min log n> <max log n>): proofs of random statements
large enough for the round-0 tables (n >= 2^17), the shared G2 chains and the fused fold of rounds 0 and 1, with a RANDOM look-ahead plan
(0 .. 48 eighths: whole and partial items, so x1 is sometimes known with x0 and sometimes late), random switches among the forms that must agree
(RIPP_FUSE_TABLES, RIPP_NO_FUSE, RIPP_NO_SHARE, RIPP_NO_XSCALE, RIPP_NO_FOLD_TABLES, RIPP_NO_PREBUILD) and identities / repeated points planted in every quarter -- each
proof against the CPU oracle's, one process, one resident job per statement proved twice (nothing prepared for one proof may leak into the next)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, orclib as o, ripp_amd as R

R.init(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 17
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 18
rng = np.random.default_rng(seed)
SW = ["RIPP_FUSE_TABLES", "RIPP_NO_FUSE", "RIPP_NO_SHARE", "RIPP_NO_XSCALE", "RIPP_NO_FOLD_TABLES", "RIPP_NO_PREBUILD"]
bad = cnt = capped = 0; t0 = time.time()
while time.time() - t0 < seconds:
    n = 1 << int(rng.integers(lo, hi + 1)); q = n // 4
    sa, sb, sr = (int(x) for x in rng.integers(1, 1 << 30, 3))
    a, b, r = R.synth_g1(sa, n), R.synth_g2(sb, n), R.synth_fr(sr, n)
    for k in range(4):                                        # degenerate rows in every quarter: identities, a zero coefficient, x P + P with equal operands
        i = k * q + int(rng.integers(0, q))
        c = rng.random()
        if c < 0.3: a[i] = 0
        elif c < 0.6: b[i] = 0
        elif c < 0.7: r[i] = 0
        else:
            jdx = (i + 2 * q) % n; a[jdx] = a[i]; b[jdx] = b[i]; r[jdx] = r[i]
    v = R.product_of_pairings_with_coeffs(a, b, r)
    rc, ep, ech = o.sipp_prove(a, b, r, v)
    job = R.SippJob(a, b, r)
    for rep in range(2):
        env = {"RIPP_LOOK_EIGHTHS": str(int(rng.integers(0, 49)))}
        for s in SW:
            if rng.random() < 0.25: env[s] = "1"
        if rng.random() < 0.3:                                 # a random device-memory cap (build round 5): the scratch is released first so that the cap binds and the tiers of job_precompute_round0 / pairs_cap are taken
            env["RIPP_MEM_CAP_BYTES"] = str(int(rng.integers(12, 80)) * 100_000_000 + R.device_bytes() * 0)
            R.release_scratch(); capped += 1
        os.environ.update(env)
        try:
            p, ch, st = job.prove(v)
        finally:
            for k in env: del os.environ[k]
        cnt += 1
        if not (rc == 0 and np.array_equal(p, ep) and np.array_equal(ch, ech)):
            bad += 1; print("MISMATCH", n, sa, sb, sr, env, flush=True)
    job.close()
print("proofs", cnt, "mismatches", bad, "(n = 2^%d .. 2^%d, random look-ahead plans and switches %s; %d proofs under a random RIPP_MEM_CAP_BYTES of 1.2 - 8 GB)" % (lo, hi, " ".join(SW), capped))
