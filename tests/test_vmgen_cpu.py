"""CPU: the lane-parallel VM's layer tables (ripp_amd/csrc/vm_programs.inc) are re-derived and re-validated:
every program is list-scheduled, slot-allocated and EVALUATED with Python integers against the plain formulas
(line double/add, dense Fp12 product, homogeneous doubling and complete addition on G1 and G2), the bounds the device arithmetic
relies on are re-checked, and the committed header must be exactly what the generator emits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


import pytest


@pytest.fixture(params=["bls12_381", "bls12_377"])
def curve(request):
    import vmgen
    vmgen.set_curve(request.param)
    yield request.param
    vmgen.set_curve("bls12_381")


def test_vm_schedules_validate_and_header_is_current(tmp_path, curve):
    """both curves: BLS12-377 has its own tables (u^2 = -5, xi = u, D-type twist with b' = 1/u carried times 5, b = 1)"""
    import vmgen
    progs = vmgen.validate()                       # asserts inside compare each schedule with the reference formulas
    assert {name for name, _ in progs} >= {"line_double", "line_add", "fp12_mul", "g1_hdbl", "g1_cadd", "g2_hdbl", "g2_cadd"}
    out = tmp_path / "vm_programs.inc"
    vmgen.emit(progs, str(out))
    committed = open(os.path.join(ROOT, vmgen.CURVE["header"])).read()
    assert out.read_text() == committed, vmgen.CURVE["header"] + " is stale: run python tools/vmgen.py"


def test_vm_layers_are_homogeneous_and_in_bounds(curve):
    """What vm.hpp::vm_run assumes about the tables: slot indices are bytes, a LIN op has at most 16 terms with |coefficient| <= 127 and a
    bias that fits 16 bits and covers its negative terms (values < 2p for program inputs and products, < 16p for unreduced LIN results),
    and a light LIN result stays below 16 p."""
    import vmgen
    for (name, G), c in vmgen.validate().items():
        assert c["nslots"] <= 255
        bound = {s: vmgen.BOUND_IN for s in c["ins"].values()}; bound[vmgen.ZERO_SLOT] = 0; bound[vmgen.DUMP_SLOT] = 0
        for kind, row in c["layers"]:
            assert kind in (vmgen.MUL, vmgen.LIN) and len(row) == G
            new = {}
            heavy_layer = kind == vmgen.LIN and any(op["heavy"] for op in row)
            for op in row:
                assert 0 <= op["dst"] < c["nslots"]
                if kind == vmgen.MUL:
                    assert all(0 <= a < c["nslots"] for a in op["a"]) and 0 <= op["neg"] < 16
                    ob = [sum((vmgen.NEG_K if (op["neg"] >> (2 * h + t)) & 1 else bound[op["a"][2 * h + t]]) for t in range(2) if op["a"][2 * h + t] != vmgen.ZERO_SLOT or t == 0) for h in range(2)]
                    assert all(bound[a] <= vmgen.LIGHT_MAX for a in op["a"]) and ob[0] * ob[1] <= vmgen.VMAX
                    new[op["dst"]] = 2
                else:
                    assert len(op["terms"]) <= vmgen.TMAX and all(abs(cf) <= vmgen.COEF_MAX and 0 <= sl < c["nslots"] for cf, sl in op["terms"]) and 0 <= op["nbias"] < 65536
                    neg = sum(-cf * bound[sl] for cf, sl in op["terms"] if cf < 0); pos = sum(cf * bound[sl] for cf, sl in op["terms"] if cf > 0)
                    assert op["nbias"] >= neg and pos + op["nbias"] <= vmgen.HEAVY_MAX
                    assert heavy_layer or pos + op["nbias"] <= vmgen.LIGHT_MAX
                    new[op["dst"]] = 2 if heavy_layer else pos + op["nbias"]
            bound.update(new)
        for s in c["outs"].values(): assert bound[s] <= vmgen.BOUND_IN          # kernels read outputs back as canonical values


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


def test_binary_gcd_inversion_model():
    """tools/inv_model.py restates the device inversion (fp_inv_bingcd) limb for limb on Python integers and asserts its invariants; the GPU
    tests compare the kernel itself with the host's Fermat inverse through every normalisation."""
    import inv_model
    assert inv_model.self_test(samples=300) == 26
