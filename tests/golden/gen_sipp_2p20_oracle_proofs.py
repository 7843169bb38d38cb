#!/usr/bin/env python3
"""Regenerates tests/golden/sipp_2p20_oracle_proofs.npz: the CPU ORACLE's SIPP proofs of the headline statement (bench.py's: a_i = (1000 + i) G1,
b_i = (2000 + i) G2, r_i = SplitMix64(0), n = 2^20) on BLS12-381 and BLS12-377 -- value, the 40 GT elements, the 20 challenges.

Why a fixture: one oracle proof at n = 2^20 costs ~80-110 s of 16 CPUs, and the GPU suite needs both curves' proofs -- 40 % of its running time.
The fixture is a CACHE of the oracle's output, not a second source of truth:
  * tests/test_gpu_full_size.py compares the GPU proofs with it AND has the oracle's verifier (which re-derives every challenge from its own
    Blake2s of the statement) accept them; RIPP_TEST_LIVE_ORACLE=1 recomputes the oracle's proofs instead of loading them;
  * bench.py's cpu_baseline leg runs the oracle's prover at n = 2^20 LIVE in every default run and compares its proof with the GPU's.
Takes ~6 min on 8 cores:  python tests/golden/gen_sipp_2p20_oracle_proofs.py"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np


CANARY_N = 1 << 10


def statement_proof(o, n):
    """The oracle's proof of bench.py's statement generators at size n: (value, proof, challenges).  ONE code path for the fixture and its canary."""
    a, b, r = o.gen_g1(1000, n), o.gen_g2(2000, n), o.gen_scalars(0, n)
    value = o.product_of_pairings_with_coeffs(a, b, r)
    rc, proof, ch = o.sipp_prove(a, b, r, value)
    assert rc == 0 and o.sipp_verify(a, b, r, value, proof) == 1
    return value, proof, ch


def main():
    """Default: everything (~6 min).  --canary-only: keep the n = 2^20 proofs of the existing file and recompute the n = 2^10 canaries only -- for use
    after an oracle edit that is KNOWN not to change any output; tests/test_oracle_cpu.py::test_full_size_fixture_is_current then says whether it did."""
    path = os.path.join(HERE, "sipp_2p20_oracle_proofs.npz")
    out = dict(np.load(path)) if "--canary-only" in sys.argv else {}
    for tag, modname in (("381", "orclib"), ("377", "orclib377")):
        o = __import__(modname)
        if "--canary-only" not in sys.argv:
            t0 = time.time()
            out["value_" + tag], out["proof_" + tag], out["ch_" + tag] = statement_proof(o, 1 << 20)
            print(f"BLS12-{tag}: oracle proof of the n = 2^20 statement in {time.time() - t0:.0f} s", flush=True)
        # the canary: the same generators and the same prover at n = 2^10, recomputed by the CPU suite in a second -- an oracle (or generator) edit that changes
        # outputs without a refresh of this file fails there, for both curves
        out["canary_value_" + tag], out["canary_proof_" + tag], out["canary_ch_" + tag] = statement_proof(o, CANARY_N)
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
