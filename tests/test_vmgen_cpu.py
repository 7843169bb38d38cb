"""CPU: the lane-parallel VM's layer tables (ripp_amd/csrc/vm_programs.inc) are re-derived and re-validated:
every program is list-scheduled, slot-allocated and EVALUATED with Python integers against the plain formulas
(line double/add, sparse and dense Fp12 products, homogeneous group law on G1 and G2), and the committed header must be
exactly what the generator emits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_vm_schedules_validate_and_header_is_current(tmp_path):
    import vmgen
    progs = vmgen.validate()                       # asserts inside compare each schedule with the reference formulas
    assert {name for name, _ in progs} >= {"line_double", "line_add", "acc_014", "fp12_mul", "g1_hdbl", "g1_hadd", "g2_hdbl", "g2_hadd"}
    out = tmp_path / "vm_programs.inc"
    vmgen.emit(progs, str(out))
    committed = open(os.path.join(ROOT, "ripp_amd", "csrc", "vm_programs.inc")).read()
    assert out.read_text() == committed, "vm_programs.inc is stale: run python tools/vmgen.py"


def test_vm_layers_are_homogeneous_and_in_bounds():
    import vmgen
    for (name, G), c in vmgen.validate().items():
        assert c["nslots"] <= 255                                  # slot indices are bytes
        for kind, row in c["layers"]:
            assert kind in (vmgen.MUL, vmgen.LIN) and len(row) == G
            for op in row:
                assert 0 <= op["dst"] < c["nslots"] and all(0 <= a < c["nslots"] for a in op["a"])
                if kind == vmgen.MUL:
                    assert op["half"] == 0 and op["sh"] == 0


def test_kaliski_fix_table():
    """KALISKI_FIX[k] = R^3 2^-k mod p (bls12_381/inv_table.inc), the constant that turns the almost-Montgomery inverse
    into the Montgomery form of a^-1; spot-check rows and the algorithm itself on integers."""
    p = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    R = 1 << 384
    rows = [l for l in open(os.path.join(ROOT, "ripp_amd", "csrc", "bls12_381", "inv_table.inc")) if l.startswith("{")]
    assert len(rows) == 769
    for k in (0, 1, 381, 500, 768):
        limbs = [int(x.rstrip("u"), 16) for x in rows[k].strip().rstrip(",").strip("{}").split(",")]
        assert sum(v << (32 * i) for i, v in enumerate(limbs)) == pow(R, 3, p) * pow(2, -k, p) % p
    import random
    rnd = random.Random(5)
    for a in [1, 2, p - 1] + [rnd.randrange(1, p) for _ in range(50)]:
        am = a * R % p
        u, v, r, s, k = p, am, 0, 1, 0
        while v > 0:
            if u % 2 == 0: u //= 2; s *= 2
            elif v % 2 == 0: v //= 2; r *= 2
            elif u > v: u = (u - v) // 2; r += s; s *= 2
            else: v = (v - u) // 2; s += r; r *= 2
            k += 1
            assert k <= 768
        r = r - p if r >= p else r
        assert (p - r) * (pow(R, 3, p) * pow(2, -k, p) % p) * pow(R, -1, p) % p == pow(a, -1, p) * R % p
