#!/usr/bin/env python3
"""Compare the golden vectors of this repository with a dump produced by REAL arkworks (rust/ripp-hip/examples/dump_kat.rs, dump_kat_377.rs).

  python3 tools/compare_kat.py tests/golden/bls12_381_vectors.json kat_arkworks.json
  python3 tools/compare_kat.py tests/golden/bls12_377_vectors.json kat_arkworks_377.json --curve bls12_377

Every key the dump holds that the golden file holds too must be equal (hex strings compared case-insensitively, integers as integers).
A `base_case` section (dump_kat_377.rs: the reference's own `prove_and_verify_base_case`, sipp/src/lib.rs:232-254, with its inputs) is
checked against this repository's CPU ORACLE instead: the dumped inputs go through oracle/'s SIPP prover and the value, seed digest, proof
and challenges must equal arkworks' (skipped with a note when the oracle library cannot be loaded).
Exit status 0 = the oracle's conventions are pinned by arkworks itself; exit status 1 lists each [ark-mem] convention that differs
(SURVEY.md section 8c)."""
import json
import os
import sys


def norm(v):
    if isinstance(v, str):
        s = v.lower()
        if s.startswith("0x"):
            return int(s, 16)
        return s
    if isinstance(v, list):
        return [norm(x) for x in v]
    return v


def check_base_case(bc, curve):
    """the dumped statement through the oracle's own SIPP prover; returns the list of differing members"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    o = __import__("orclib377" if curve == "bls12_377" else "orclib")
    o.lib()
    a = o.g1_array([None if p is None else (int(p[0], 16), int(p[1], 16)) for p in bc["a"]])
    b = o.g2_array([None if q is None else ((int(q[0][0], 16), int(q[0][1], 16)), (int(q[1][0], 16), int(q[1][1], 16))) for q in bc["b"]])
    r = o.fr_array([int(x, 16) for x in bc["r"]])
    bad = []
    value = o.product_of_pairings_with_coeffs(a, b, r)
    if o.ser_gt(value).hex() != bc["value"].lower():
        bad.append("base_case.value")
    if o.sipp_seed_digest(a, b, r, value).hex() != bc["seed_digest"].lower():
        bad.append("base_case.seed_digest")
    rc, proof, ch = o.sipp_prove(a, b, r, value)
    got = [[o.ser_gt(proof[2 * j]).hex(), o.ser_gt(proof[2 * j + 1]).hex()] for j in range(len(proof) // 2)]
    if rc != 0 or got != norm(bc["proof"]):
        bad.append("base_case.proof")
    if [o.limbs_to_fr(c) for c in ch] != norm(bc["challenges"]):
        bad.append("base_case.challenges")
    return bad


def main(argv):
    args = [a for a in argv[1:] if not a.startswith("--")]
    curve = "bls12_377" if "bls12_377" in " ".join(argv[1:]) else "bls12_381"
    if "--curve" in argv:
        curve = argv[argv.index("--curve") + 1]; args = [a for a in args if a != curve]
    gold, ark = json.load(open(args[0])), json.load(open(args[1]))
    bad, checked = [], 0
    for sec, vals in ark.items():
        if sec == "base_case":
            try:
                bad += check_base_case(vals, curve); checked += 4
            except OSError as ex:
                print(f"  (base_case not checked: the CPU oracle could not be loaded: {ex})")
            continue
        for k, v in vals.items():
            if sec not in gold or k not in gold[sec]:
                print(f"  (extra in dump, not in golden: {sec}.{k})"); continue
            checked += 1
            if norm(gold[sec][k]) != norm(v):
                bad.append(f"{sec}.{k}")
    if bad:
        print("MISMATCH against arkworks:", ", ".join(bad)); return 1
    print("all", checked, "known answers equal arkworks' -- oracle pinned")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
