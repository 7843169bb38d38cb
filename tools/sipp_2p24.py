#!/usr/bin/env python3
"""A real out-of-memory proof: SIPP at n = 2^24 (16 x the headline; the reference's harness takes any <log_max>, sipp/examples/scaling-ipp.rs:22-32,57-62)
with NO mem_cap_bytes.  The three-quarter fold tables alone would be 229 GB, so hipMemGetInfo itself must drive the call down the memory tiers
(ripp_stats.mem_tier > 0).  Prove -> the CPU oracle's verifier accepts (it re-derives every challenge from its own Blake2s of the 5.4 GB statement) ->
ripp_release_scratch returns the library to its baseline.  Writes a small report (default profiles/r06_sipp_2p24.txt).
    python tools/sipp_2p24.py [--log-n 24] [--out profiles/r06_sipp_2p24.txt] [--no-oracle]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_sipp_2p24.txt"))
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    import ripp_amd as R
    R.init(0)
    n = 1 << args.log_n
    lines = []

    def say(msg):
        print(msg, flush=True); lines.append(msg)
    free0, total = torch.cuda.mem_get_info(0)
    base = R.device_bytes()
    say(f"# SIPP prove at n = 2^{args.log_n} on one MI355X, no mem_cap_bytes: the device's free memory decides the tier")
    say(f"device memory: {total / 2**30:.1f} GB total, {free0 / 2**30:.1f} GB free before the statement; library holds {base} bytes")
    t0 = time.time()
    a, b, r = R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n)
    say(f"statement: {(a.nbytes + b.nbytes + r.nbytes) / 2**30:.2f} GB on the host (generated on the device in {time.time() - t0:.1f} s)")
    t0 = time.time(); value = R.product_of_pairings_with_coeffs(a, b, r); say(f"claimed value (ripp_pairing_product_coeffs_a): {time.time() - t0:.2f} s")
    for k in range(2):
        t0 = time.time(); proof, ch, st = R.SIPP.prove_one_shot(a, b, r, value); dt = time.time() - t0
        say(f"proof {k}: {dt:.3f} s = {n / dt / 1e6:.3f} M pairs/s; mem_tier {int(st['mem_tier'])} (bits 0-2: fold-table tier, 8: line buffer cut, 16: split-form table round); "
            f"library holds {int(st['device_bytes']) / 2**30:.1f} GB; statement hash {st['statement_hash_ms']:.0f} ms, look-ahead items {int(st['look_items'])}")
    assert proof.shape == (2 * args.log_n, 72)
    assert int(st["mem_tier"]) > 0 or args.log_n < 24, "the full tier at this size cannot fit the device: mem_tier must be > 0"
    t0 = time.time(); ok = R.SIPP.verify(a, b, r, value, proof); say(f"engine verifier: {'accepts' if ok else 'REJECTS'} ({time.time() - t0:.1f} s)")
    assert ok
    if not args.no_oracle:
        import orclib as o
        t0 = time.time(); verdict = o.sipp_verify(a, b, r, value, proof)
        say(f"CPU oracle verifier ({o.effective_cpus()} threads): {'accepts' if verdict == 1 else 'REJECTS'} ({time.time() - t0:.0f} s)")
        assert verdict == 1
        bad = proof.copy(); bad[5, 7] ^= 1
        t0 = time.time(); verdict = o.sipp_verify(a, b, r, value, bad)
        say(f"CPU oracle verifier on the proof with one bit flipped: {'accepts (!)' if verdict == 1 else 'rejects'} ({time.time() - t0:.0f} s)")
        assert verdict == 0
    R.release_scratch()
    left = R.device_bytes()
    say(f"after ripp_release_scratch: library holds {left} bytes (baseline {base}); device free {torch.cuda.mem_get_info(0)[0] / 2**30:.1f} GB")
    assert left <= base + (1 << 16)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
