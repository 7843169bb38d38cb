"""CPU: the C-ABI library loads, exports every symbol include/ripp_hip.h declares, fails loudly without a device
(no CPU fallback), and its device-free host helpers (final exponentiation, serialisation, Fiat-Shamir step) agree
with the oracle.  No compute entry point is exercised here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    from ripp_amd._lib import lib
    return lib()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ripp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ripp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(hiplib):
    names = declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(hiplib, n)]
    assert not missing, f"declared in include/ripp_hip.h but not exported: {missing}"


def test_bls12_377_library_exports_the_same_abi_and_its_own_point_encodings():
    """libripp_hip_377.so is the same engine: every declared symbol, the proof-struct wire format included -- with ark-ec's GENERIC
    short-Weierstrass encodings (flags in the LAST byte; compressed = x alone) instead of ark-bls12-381's zcash layout.  The compressed images
    of the generators, their negatives and the identities equal the model's (tests/golden/bls12_377_vectors.json); device-free."""
    import json
    import ripp_amd.bls12_377 as R7
    import orclib377 as o7
    L = R7.lib()
    missing = [n for n in declared_symbols() if not hasattr(L, n)]
    assert not missing, missing
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "bls12_377_vectors.json")))["generators"]
    g1 = o7.g1_array([(int(g["g1"][0], 16), int(g["g1"][1], 16))]); g2 = o7.g2_array([((int(g["g2"][0][0], 16), int(g["g2"][0][1], 16)), (int(g["g2"][1][0], 16), int(g["g2"][1][1], 16)))])
    n1 = g1.copy(); n1[0, 6:] = o7.fp_to_limbs((o7.P - o7.limbs_to_fp(g1[0, 6:])) % o7.P)
    n2 = g2.copy()
    for k in (12, 18): n2[0, k:k + 6] = o7.fp_to_limbs((o7.P - o7.limbs_to_fp(g2[0, k:k + 6])) % o7.P)
    assert R7.ser_g1_compressed(g1[0]).hex() == g["ser_g1_compressed"] and R7.ser_g2_compressed(g2[0]).hex() == g["ser_g2_compressed"]
    assert R7.ser_g1_compressed(n1[0]).hex() == g["ser_g1_neg_compressed"] and R7.ser_g2_compressed(n2[0]).hex() == g["ser_g2_neg_compressed"]
    assert R7.ser_g1_compressed(np.zeros(12, dtype=np.uint64)).hex() == g["ser_g1_inf_compressed"] and R7.ser_g2_compressed(np.zeros(24, dtype=np.uint64)).hex() == g["ser_g2_inf_compressed"]


def test_abi_guard_and_config_defaults_need_no_device(hiplib):
    """The ABI guard (version + sizeof(ripp_stats), checked by the loader) and ripp_config: the defaults are readable without a device, a
    struct of another size is rejected, ripp_configure(NULL) returns to the defaults."""
    import ripp_amd as R
    from ripp_amd._lib import RippConfig, RippStats, RIPP_ABI_VERSION
    assert hiplib.ripp_abi_version() == RIPP_ABI_VERSION and hiplib.ripp_stats_size() == ctypes.sizeof(RippStats)
    c = R.config_default()
    assert c.struct_size == ctypes.sizeof(RippConfig) and c.look_eighths == -1 and c.ranks_per_device == 1
    assert c.tail_pipe_max == 1 << 11 and c.fold_tab_min == 32768 and c.no_vm == 0 and c.no_precompute == 0 and c.lp_fq_min == 0
    assert c.msm_chunk_min == 1 << 20 and c.msm_lds_sort_min == 0 and c.no_prebuild == 0 and c.fq_min_g1 == 1 << 12       # (members added in build round 4: the layout of the binding follows the header)
    assert c.mem_cap_bytes == 0 and c.hot_workers == 0 and c.no_job_cache == 0                                              # (build round 5, ABI version 6)
    assert c.no_lp_karatsuba == 0 and c.comm_timeout_ms == 0 and c.plan_derate_pct == 0 and c.n_devices == 0                 # (build round 6, ABI version 7)
    assert ctypes.sizeof(RippConfig) == 4 * 22 + 8 * 16 + 4 * 2 + 4 * 4 and ctypes.sizeof(RippStats) == 24 * 8
    assert hiplib.ripp_device_bytes() == 0                                                                                  # nothing allocated before the first device call
    bad = RippConfig(); bad.struct_size = 8
    assert hiplib.ripp_configure(ctypes.byref(bad)) == 4             # RIPP_ERR_ARG
    assert hiplib.ripp_configure(ctypes.byref(c)) == 0 and hiplib.ripp_configure(None) == 0
    with pytest.raises(AttributeError):
        R.configure(no_such_member=1)
    R.configure()


def test_no_cpu_fallback_without_device(hiplib):
    if hiplib.ripp_device_count() > 0:
        pytest.skip("a HIP device is present; the refusal path is exercised on the CPU-only builder")
    a = np.zeros((2, 12), dtype=np.uint64); b = np.zeros((2, 24), dtype=np.uint64); out = np.zeros(72, dtype=np.uint64)
    rc = hiplib.ripp_pairing_product_a(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(2), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 3                                                   # RIPP_ERR_DEVICE
    import ripp_amd as R
    with pytest.raises(R.DeviceError):
        R.product_of_pairings(a, b)
    with pytest.raises(R.DeviceError):
        R.init(0)


def test_length_error_is_reported_before_touching_the_device(hiplib):
    import ripp_amd as R
    with pytest.raises(R.InnerProductError) as ei:
        R.PairingInnerProduct.inner_product(np.zeros((5, 18), dtype=np.uint64), np.zeros((4, 36), dtype=np.uint64))
    assert (ei.value.left, ei.value.right) == (5, 4) and str(ei.value) == "left length, right length: 5, 4"
    with pytest.raises(R.InnerProductError):
        R.MultiexponentiationInnerProductG1.inner_product(np.zeros((3, 18), dtype=np.uint64), np.zeros((2, 4), dtype=np.uint64))


def test_sipp_slice_length_mismatch_never_reaches_the_library(hiplib):
    """SIPP.prove / SIPP.verify / product_of_pairings_with_coeffs check len(a) == len(b) == len(r) (and an even proof length) on the
    Python side: the C entry points take ONE n and would read past a short slice (sipp/src/lib.rs:48-49, 116-117)."""
    import ripp_amd as R
    z = lambda n, c: np.zeros((n, c), dtype=np.uint64)
    gt = np.zeros(72, dtype=np.uint64)
    for na, nb, nr in ((4, 2, 4), (4, 4, 2), (2, 4, 4)):
        with pytest.raises(AssertionError):
            R.SIPP.prove(z(na, 12), z(nb, 24), z(nr, 4), gt)
        with pytest.raises(AssertionError):
            R.SIPP.verify(z(na, 12), z(nb, 24), z(nr, 4), gt, np.zeros((4, 72), dtype=np.uint64))
        with pytest.raises(AssertionError):
            R.product_of_pairings_with_coeffs(z(na, 12), z(nb, 24), z(nr, 4))
    with pytest.raises(ValueError):
        R.SIPP.verify(z(4, 12), z(4, 24), z(4, 4), gt, np.zeros((3, 72), dtype=np.uint64))
    with pytest.raises(AssertionError):
        R.SippJob(z(4, 12), z(4, 24), z(2, 4))


def test_host_helpers_match_oracle(hiplib, orc, vectors):
    import ripp_amd as R
    a, b = orc.gen_g1(77, 3), orc.gen_g2(88, 3)
    ml = orc.miller_product_a(a, b)
    assert np.array_equal(R.final_exponentiation(ml), orc.final_exp(ml))
    e = orc.pairing_product_a(a, b)
    assert R.ser_gt(e) == orc.ser_gt(e) and R.ser_g1(a[1]) == orc.ser_g1(a[1]) and R.ser_g2(b[2]) == orc.ser_g2(b[2])
    assert R.ser_g1(np.zeros(12, dtype=np.uint64)) == orc.ser_g1(orc.u64(12))
    s = orc.gen_scalars(3, 1)[0]
    assert R.ser_fr(s) == orc.ser_fr(s)
    assert np.array_equal(R.gt_mul(e, ml), orc.gt_mul(e, ml))
    # statement digest (sipp/src/lib.rs:56-59) incl. the multi-block path
    n = 40
    A, B, r = orc.gen_g1(5, n), orc.gen_g2(6, n), orc.gen_scalars(7, n)
    assert R.sipp_seed_digest(A, B, r, e) == orc.sipp_seed_digest(A, B, r, e)
    v = vectors["sipp4"]
    from helpers import g1arr, g2arr, frarr, gt_from_bytes
    assert R.sipp_seed_digest(g1arr(v["a"]), g2arr(v["b"]), frarr(v["r"]), gt_from_bytes(v["value"])).hex() == v["seed_digest"]


def test_range_split_final_exponentiation_is_the_same_value(hiplib, orc):
    """Between two kernels the provers turn 68 per-step products into final_exponentiation(miller_combine(.)) in bit RANGES on host workers
    (engine.hip pairing_values): for every number of ranges the value must be the one the plain composition gives."""
    import ctypes
    rows = np.stack([orc.miller_product_a(orc.gen_g1(10 + k, 2), orc.gen_g2(20 + k, 2)) for k in range(2 * 68)]).astype(np.uint64)      # arbitrary invertible Fp12 values
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    exp = []
    for k in range(2):
        m = np.zeros(72, dtype=np.uint64); z = np.zeros(72, dtype=np.uint64)
        assert hiplib.ripp_miller_combine(p(np.ascontiguousarray(rows[68 * k:68 * (k + 1)])), p(m)) == 0 and hiplib.ripp_final_exp(p(m), p(z)) == 0
        exp.append(z)
    for parts in (0, 1, 2, 3, 4, 7, 63):
        out = np.zeros((2, 72), dtype=np.uint64)
        assert hiplib.ripp_pairing_values(p(rows), 2, parts, p(out)) == 0
        assert np.array_equal(out[0], exp[0]) and np.array_equal(out[1], exp[1]), parts
    assert np.array_equal(exp[0], orc.final_exp(orc.miller_combine(rows[:68]))) if hasattr(orc, "miller_combine") else True


def test_fiat_shamir_step_matches_golden(hiplib, vectors):
    """ripp_sipp_challenge = absorb (z_l, z_r) then draw x (sipp/src/lib.rs:80-85) against the model's SIPP transcript."""
    from helpers import gt_from_bytes
    v = vectors["sipp4"]
    seed = (ctypes.c_uint8 * 32).from_buffer_copy(bytes.fromhex(v["seed_digest"]))
    for j, (zl, zr) in enumerate(v["proof"]):
        x = np.zeros(4, dtype=np.uint64)
        zl_, zr_ = gt_from_bytes(zl), gt_from_bytes(zr)
        rc = hiplib.ripp_sipp_challenge(seed, zl_.ctypes.data_as(ctypes.c_void_p), zr_.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        import orclib
        assert hex(orclib.limbs_to_fr(x)) == v["challenges"][j]


def _build_c_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"),
                           "-L" + os.path.join(ROOT, "ripp_amd", "lib"), "-lripp_hip", "-Wl,-rpath," + os.path.join(ROOT, "ripp_amd", "lib"), "-o", exe])
    return exe


def test_header_is_plain_c_and_demo_fails_loudly_without_device(tmp_path):
    """include/ripp_hip.h must be consumable from C99 (the drop-in boundary has no C++ in it); without a GPU the demo stops with the
    'no CPU fallback' message instead of computing anything."""
    import subprocess
    exe = _build_c_demo(tmp_path)
    from ripp_amd._lib import lib
    if lib().ripp_device_count() > 0:
        pytest.skip("a HIP device is present; the GPU variant of this test covers the run")
    p = subprocess.run([exe, "4"], capture_output=True, text=True)
    assert p.returncode == 2 and "no CPU fallback" in p.stderr


def test_library_point_encoding_matches_public_literals(hiplib, orc):
    """The library's own (host-side) serialisers against the published zcash / IETF images of the BLS12-381 generators -- literals, not
    model output (tests/test_oracle_cpu.py holds the citations)."""
    import ripp_amd as R
    from test_oracle_cpu import G1_COMPRESSED, G2_COMPRESSED, G1_X, G1_Y, G2_X0, G2_X1, G2_Y0, G2_Y1
    g1, g2 = orc.gen_g1(1, 1)[0], orc.gen_g2(1, 1)[0]
    assert R.ser_g1_compressed(g1).hex() == G1_COMPRESSED and R.ser_g2_compressed(g2).hex() == G2_COMPRESSED
    assert R.ser_g1(g1).hex() == "%096x%096x" % (G1_X, G1_Y)
    assert R.ser_g2(g2).hex() == "%096x%096x%096x%096x" % (G2_X1, G2_X0, G2_Y1, G2_Y0)
    assert R.ser_g1_compressed(np.zeros(12, dtype=np.uint64)).hex() == "c0" + "00" * 47      # infinity: compression + infinity bits


def test_blake2s_matches_hashlib_at_every_length_class(hiplib):
    """The statement hash's Blake2s (host_fs.hpp + the generated x86-64 bulk loop blake2s_x64.S, which handles n / 64 - 1 blocks of a call and stages
    64 bytes ahead) against hashlib.blake2s: every length 0..700 -- below, at and above the 192-byte threshold of the assembly path, every residue
    mod 64 -- a few large odd sizes, unaligned starts, and a buffer that ends exactly at the end of a page-aligned mapping (an overread would fault)."""
    import ctypes.util
    import hashlib
    import mmap
    rng = np.random.default_rng(7)
    data = rng.integers(0, 256, size=(1 << 20) + 777, dtype=np.uint8)
    out = np.zeros(32, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for n in list(range(0, 701)) + [4095, 4096, 4097, 65536 + 13, (1 << 20) + 777]:
        for off in ((0, 1, 7) if n < 800 else (0, 3)):
            if off + n > len(data):
                continue
            view = data[off:off + n]
            assert hiplib.ripp_blake2s(p(view) if n else None, ctypes.c_size_t(n), p(out)) == 0
            assert bytes(out) == hashlib.blake2s(view.tobytes()).digest(), (n, off)
    # the buffer's last byte is the last byte of a mapping followed by an unmapped page
    mm = mmap.mmap(-1, 2 * mmap.PAGESIZE)
    buf = (ctypes.c_uint8 * (2 * mmap.PAGESIZE)).from_buffer(mm)
    base = ctypes.addressof(buf)
    libc = ctypes.CDLL(ctypes.util.find_library("c"), use_errno=True)
    assert libc.mprotect(ctypes.c_void_p(base + mmap.PAGESIZE), ctypes.c_size_t(mmap.PAGESIZE), 0) == 0          # PROT_NONE on the second page
    for n in (192, 200, 256, 1000, mmap.PAGESIZE):
        src = rng.integers(0, 256, size=n, dtype=np.uint8)
        ctypes.memmove(base + mmap.PAGESIZE - n, src.ctypes.data, n)
        assert hiplib.ripp_blake2s(ctypes.c_void_p(base + mmap.PAGESIZE - n), ctypes.c_size_t(n), p(out)) == 0
        assert bytes(out) == hashlib.blake2s(src.tobytes()).digest(), n
    libc.mprotect(ctypes.c_void_p(base + mmap.PAGESIZE), ctypes.c_size_t(mmap.PAGESIZE), 3)
    del buf; mm.close()


def test_blake2s_assembly_is_the_generators_output():
    """ripp_amd/csrc/blake2s_x64.S is generated (tools/ubench/gen_blake2s_x64.py); its first line records the arguments: regenerating must reproduce it."""
    import subprocess
    import sys
    path = os.path.join(ROOT, "ripp_amd", "csrc", "blake2s_x64.S")
    text = open(path).read()
    first = text.splitlines()[0]
    assert first.startswith("# GENERATED by tools/ubench/gen_blake2s_x64.py ")
    args = first.split("gen_blake2s_x64.py ", 1)[1].split(" -- ")[0].split()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ubench", "gen_blake2s_x64.py")] + args, capture_output=True, text=True, check=True).stdout
    assert out == text, "ripp_amd/csrc/blake2s_x64.S is stale: regenerate it with the arguments on its first line"
