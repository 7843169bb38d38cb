"""CPU: tools/compare_kat.py -- the checker of the arkworks hand-off (rust/ripp-hip/examples/dump_kat.rs, dump_kat_377.rs) -- accepts a dump
that equals the golden file, exits 1 on a doctored one and names the member, and checks a `base_case` section (the reference's own SIPP test
with its inputs, sipp/src/lib.rs:232-254) by running the dumped statement through the CPU oracle."""
import copy
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "compare_kat.py")


def run(gold, dump, tmp_path, *extra):
    p = tmp_path / "dump.json"; p.write_text(json.dumps(dump))
    r = subprocess.run([sys.executable, TOOL, gold, str(p), *extra], capture_output=True, text=True, timeout=600)
    return r.returncode, r.stdout + r.stderr


OUTPUT_KEYS = {"generators": None, "pairing_generators": None, "bilinearity": ("gt", "gt_pow"), "product8": ("gt",), "msm8": ("g1", "g2"), "fold": ("g1", "g2"),
               "sipp4": ("value", "seed_digest", "proof", "challenges"), "primitives": ("blake2s_abc", "blake2b_abc"), "fsrng": ("u128_0", "u128_after_absorb"),
               "tipa4": ("com", "proof_uncompressed", "proof_compressed"),
               "aggregate4": ("r", "com_a", "com_b", "com_c", "ip_ab", "agg_c", "tipa_proof_ab", "tipa_proof_c")}


def as_dump(gold):
    """what a faithful arkworks run would print: the OUTPUT members of every section the golden file holds"""
    return {sec: {k: v for k, v in vals.items() if OUTPUT_KEYS.get(sec) is None or k in OUTPUT_KEYS[sec]} for sec, vals in gold.items() if sec in OUTPUT_KEYS}


@pytest.mark.parametrize("name", ["bls12_381_vectors.json", "bls12_377_vectors.json"])
def test_faithful_dump_is_accepted_and_a_doctored_one_is_not(tmp_path, name):
    path = os.path.join(ROOT, "tests", "golden", name)
    gold = json.load(open(path))
    dump = as_dump(gold)
    rc, out = run(path, dump, tmp_path)
    assert rc == 0 and "oracle pinned" in out, out
    # every section the hand-off promises is really compared (dump_kat.rs prints aggregate4; both dumpers print sipp4)
    if name.startswith("bls12_381"):
        assert "aggregate4" in dump and set(dump["aggregate4"]) == set(OUTPUT_KEYS["aggregate4"])
    bad = copy.deepcopy(dump)
    z = bad["sipp4"]["proof"][1][0]; bad["sipp4"]["proof"][1][0] = z[:-2] + ("00" if z[-2:] != "00" else "01")
    rc, out = run(path, bad, tmp_path)
    assert rc == 1 and "sipp4.proof" in out, out
    if "aggregate4" in dump:
        bad = copy.deepcopy(dump); bad["aggregate4"]["r"] = hex(int(bad["aggregate4"]["r"], 16) ^ 1)
        rc, out = run(path, bad, tmp_path)
        assert rc == 1 and "aggregate4.r" in out, out


def test_base_case_section_goes_through_the_oracle(tmp_path):
    """a `base_case` as dump_kat_377.rs prints it (here: made by the oracle itself on a statement with an identity) passes; with one byte of
    the value or one challenge changed the tool exits 1 and names the member"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib377 as o
    o.lib()
    n = 8
    a, b, r = o.gen_g1(5, n), o.gen_g2(6, n), o.gen_scalars(7, n)
    a[3] = 0
    value = o.product_of_pairings_with_coeffs(a, b, r)
    rc, proof, ch = o.sipp_prove(a, b, r, value)
    assert rc == 0
    h = hex
    pg1 = lambda row: None if o.g1_from_row(row) is None else [h(c) for c in o.g1_from_row(row)]
    pg2 = lambda row: None if o.g2_from_row(row) is None else [[h(c) for c in pr] for pr in o.g2_from_row(row)]
    bc = {"a": [pg1(x) for x in a], "b": [pg2(x) for x in b], "r": [h(o.limbs_to_fr(x)) for x in r], "value": o.ser_gt(value).hex(),
          "seed_digest": o.sipp_seed_digest(a, b, r, value).hex(), "proof": [[o.ser_gt(proof[2 * j]).hex(), o.ser_gt(proof[2 * j + 1]).hex()] for j in range(3)],
          "challenges": [h(o.limbs_to_fr(c)) for c in ch]}
    gold = os.path.join(ROOT, "tests", "golden", "bls12_377_vectors.json")
    rc, out = run(gold, {"base_case": bc}, tmp_path, "--curve", "bls12_377")
    assert rc == 0, out
    bad = copy.deepcopy(bc); bad["challenges"][2] = hex(int(bad["challenges"][2], 16) + 1)
    rc, out = run(gold, {"base_case": bad}, tmp_path, "--curve", "bls12_377")
    assert rc == 1 and "base_case.challenges" in out, out
    bad = copy.deepcopy(bc); bad["value"] = ("01" if bc["value"][:2] != "01" else "02") + bc["value"][2:]
    rc, out = run(gold, {"base_case": bad}, tmp_path, "--curve", "bls12_377")
    assert rc == 1 and "base_case.value" in out, out
