#!/usr/bin/env python3
"""Counterpart of the reference's `benches/benches/poly_commit.rs` (its measurement harness for the polynomial-commitment
applications): for degree = 4^(i+1) - 1, i < num_data_points, time setup / commit / open / verify of the three schemes and print the
reference's CSV columns (poly_commit.rs:43-47: trial, scheme, function, degree, time[ms]) plus a backend column.

  python tools/poly_commit_bench.py <num_trials> <num_data_points> [--cpu-max DEGREE]

The device rows call ripp_amd.poly_commit (libripp_hip.so); with --cpu-max the oracle-backed restatement (tests/model/poly_commit_oracle.py)
is timed beside them up to that degree and every commitment / verdict is cross-checked.  Times include this module's host-side integer <->
Montgomery conversions of the coefficients (a Rust host hands field elements over as they are)."""
import argparse, csv, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "model"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("num_trials", type=int); ap.add_argument("num_data_points", type=int)
    ap.add_argument("--cpu-max", type=int, default=0, help="largest degree the CPU oracle rows are produced for (0 = none)")
    args = ap.parse_args()
    import numpy as np
    import ripp_amd as R, ripp_amd.poly_commit as P
    R.init(0)
    w = csv.writer(sys.stdout); w.writerow(["trial", "scheme", "function", "degree", "time", "backend"])
    def row(trial, scheme, fn, degree, t, backend="mi355x-hip"): w.writerow([trial, scheme, fn, degree, "%.3f" % (t * 1e3), backend]); sys.stdout.flush()
    def timed(f):
        t = time.perf_counter(); r = f(); return time.perf_counter() - t, r
    for degree in (4 ** (i + 1) - 1 for i in range(args.num_data_points)):
        rng = random.Random(0)
        alpha, beta = rng.randrange(1, P.R_MOD), rng.randrange(1, P.R_MOD)
        cpu = degree <= args.cpu_max
        if cpu:
            import orclib as o, poly_commit_oracle as PC
        # ---- KZG (poly_commit.rs:51-129)
        t, (powers, v_srs) = timed(lambda: P.KZG.setup(alpha, beta, degree)); row(1, "kzg", "setup", degree, t)
        for i in range(1, args.num_trials + 1):
            p = [rng.randrange(P.R_MOD) for _ in range(degree + 1)]; z = rng.randrange(P.R_MOD); ev = P.evaluate(p, z)
            t, com = timed(lambda: P.KZG.commit(powers, p)); row(i, "kzg", "commit", degree, t)
            t, proof = timed(lambda: P.KZG.open(powers, p, z)); row(i, "kzg", "open", degree, t)
            t, ok = timed(lambda: all(P.KZG.verify(v_srs, com, z, ev, proof) for _ in range(5))); assert ok; row(i, "kzg", "verify", degree, t / 5)
            if cpu:
                epowers, ev_srs = PC.kzg_setup(alpha, beta, degree)
                t, ecom = timed(lambda: PC.kzg_commit(epowers, p)); row(i, "kzg", "commit", degree, t, "cpu-oracle")
                t, eproof = timed(lambda: PC.kzg_open(epowers, p, z)); row(i, "kzg", "open", degree, t, "cpu-oracle")
                t, ok = timed(lambda: PC.kzg_verify(ev_srs, com, z, ev, proof)); assert ok; row(i, "kzg", "verify", degree, t, "cpu-oracle")
                assert np.array_equal(o.g1_to_affine(ecom), o.g1_to_affine(com)) and np.array_equal(o.g1_to_affine(eproof), o.g1_to_affine(proof))
        # ---- IPA: the pairing-based univariate scheme (poly_commit.rs:131-203)
        U = P.UnivariatePolynomialCommitment
        t, srs = timed(lambda: U.setup(alpha, beta, degree)); row(1, "ipa", "setup", degree, t)
        v = srs[0].get_verifier_key()
        for i in range(1, args.num_trials + 1):
            p = [rng.randrange(P.R_MOD) for _ in range(degree + 1)]; z = rng.randrange(P.R_MOD); ev = P.evaluate(p, z)
            t, (com, coms) = timed(lambda: U.commit(srs, p)); row(i, "ipa", "commit", degree, t)
            t, proof = timed(lambda: U.open(srs, p, coms, z)); row(i, "ipa", "open", degree, t)
            t, ok = timed(lambda: all(U.verify(v, degree, com, z, ev, proof) for _ in range(5))); assert ok; row(i, "ipa", "verify", degree, t / 5)
            if cpu:
                xd, yd = U.bivariate_degrees(degree); s = PC.bi_setup(alpha, beta, xd, yd); ys = PC.split(p, xd, yd); pt = (pow(z, yd + 1, P.R_MOD), z)
                t, (ecom, ecoms) = timed(lambda: PC.bi_commit(s, ys)); row(i, "ipa", "commit", degree, t, "cpu-oracle")
                t, eproof = timed(lambda: PC.bi_open(s, ys, ecoms, pt)); row(i, "ipa", "open", degree, t, "cpu-oracle")
                t, ok = timed(lambda: PC.bi_verify(s["v"], com, pt, ev, proof)); assert ok; row(i, "ipa", "verify", degree, t, "cpu-oracle")
                assert np.array_equal(ecom, com) and np.array_equal(eproof["ip_proof"]["tr"], proof["ip_proof"]["tr"])
        srs[0].close()
        # ---- transparent IPA (poly_commit.rs:205-277)
        T = P.transparent.UnivariatePolynomialCommitment
        t, ck = timed(lambda: T.setup(700, 900, degree)); row(1, "transparent_ipa", "setup", degree, t)
        for i in range(1, args.num_trials + 1):
            p = [rng.randrange(P.R_MOD) for _ in range(degree + 1)]; z = rng.randrange(P.R_MOD); ev = P.evaluate(p, z)
            t, (com, coms) = timed(lambda: T.commit(ck, p)); row(i, "transparent_ipa", "commit", degree, t)
            t, proof = timed(lambda: T.open(ck, p, coms, z)); row(i, "transparent_ipa", "open", degree, t)
            t, ok = timed(lambda: T.verify(ck, com, z, ev, proof)); assert ok; row(i, "transparent_ipa", "verify", degree, t)


if __name__ == "__main__":
    main()
