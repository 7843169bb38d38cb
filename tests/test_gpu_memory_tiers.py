"""Memory-aware degradation (ripp_config.mem_cap_bytes; engine.hip: Engine::mem_fits / pairs_cap, job_precompute_round0's tiers) at n = 2^17,
where every tier is reachable with a cap: the proof must be the oracle's in EVERY tier, the tier the call took is reported in
ripp_stats.mem_tier, and after ripp_release_scratch the library is back to what it held before.  One proof at n = 2^22 -- four times the
headline size, the reference's harness takes any <log_max> (sipp/examples/scaling-ipp.rs:22-32,57-62) -- is checked by the ORACLE's verifier."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_capped_proofs_equal_the_oracle_in_every_tier(engine, orc):
    R = engine
    n = 1 << 17
    a, b, r = R.synth_g1(321, n), R.synth_g2(654, n), R.synth_fr(9, n)
    value = R.product_of_pairings_with_coeffs(a, b, r)
    rc, eproof, ech = orc.sipp_prove(a, b, r, value)
    assert rc == 0
    # the look-ahead plan is forced (both values of round 1: the plan of the headline size) so that the fused three-quarter tables are the starting tier
    R.configure(look_eighths=16)
    R.release_scratch()
    base = R.device_bytes()
    proof, ch, st = R.SIPP.prove_one_shot(a, b, r, value)
    assert np.array_equal(proof, eproof) and np.array_equal(ch, ech) and st["mem_tier"] == 0
    full = int(st["device_bytes"])
    assert full > base + (1 << 30)                                   # line buffer + tables of a 2^17 proof: > 1 GB
    seen, smallest_ok = {}, None
    cap = full
    while cap > base + (64 << 20):
        cap = base + int((cap - base) * 0.88)
        R.release_scratch()
        R.configure(look_eighths=16, mem_cap_bytes=cap)
        try:
            proof, ch, st = R.SIPP.prove_one_shot(a, b, r, value)
        except R.DeviceError:
            break                                                    # below what the statement itself needs: a clean error, never a wrong proof
        assert np.array_equal(proof, eproof) and np.array_equal(ch, ech), f"cap {cap}: the proof differs from the oracle's (tier {st['mem_tier']})"
        seen.setdefault(int(st["mem_tier"]), cap)
        if int(st["device_bytes"]) > cap:
            break                                                    # the cap is below what the statement and the smallest line buffer need: the optional structures are all gone
        smallest_ok = cap
    R.configure(); R.release_scratch()
    tiers = {t & 7 for t in seen}
    assert {1, 2}.issubset(tiers) or {1, 3}.issubset(tiers), f"tiers seen (mem_tier -> first cap): {seen}"
    assert any(t & 8 for t in seen), f"the line buffer was never cut: {seen}"
    print(f"memory tiers at n = 2^17: uncapped {full >> 20} MB held; mem_tier -> first cap (MB): {({t: c >> 20 for t, c in seen.items()})}; smallest cap honoured {smallest_ok >> 20} MB")
    assert smallest_ok is not None and smallest_ok < base + (full - base) * 6 // 10, (smallest_ok, base, full, seen)
    assert R.device_bytes() <= base + (1 << 16), "ripp_release_scratch left device memory behind"          # (a few counters of a few bytes stay)
    # and without a cap the next proof is back on the full tier
    proof, ch, st = R.SIPP.prove_one_shot(a, b, r, value)
    assert np.array_equal(proof, eproof) and st["mem_tier"] == 0


def test_sipp_prove_2p22_accepted_by_the_oracle_verifier(engine, orc):
    """n = 2^22: 4 x the headline statement (1.4 GB of statement, ~57 GB of fold tables).  The proof is checked by the ORACLE's verifier
    (sipp/src/lib.rs:109-180 restated: it re-derives every challenge from its own Blake2s of the statement and folds a, b with two 2^22-term
    MSMs) and by the engine's -- on WHATEVER memory tier the device's free memory allows (a busy device degrades, it does not fail).  Then the same proof
    again beside a dummy allocation that leaves the library ~60 GB: hipMemGetInfo itself (no mem_cap_bytes) must push the call down the tiers, and the
    proof bytes must not change.  ripp_release_scratch returns the device memory."""
    import ctypes
    # the HIP runtime libripp_hip.so is bound to (a process may map two: the ROCm installation's and the copy a PyTorch wheel bundles -- by soname one gets whichever came
    # first, and the other one has no device context): take the mapped path that is not torch's
    maps = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
    paths = sorted(set(maps), key=lambda q: ("torch" in q, q))
    assert paths, "libamdhip64 is not mapped?"
    hip = ctypes.CDLL(paths[0])
    R = engine
    n = 1 << 22
    R.release_scratch()
    base = R.device_bytes()
    a, b, r = R.synth_g1(1000, n), R.synth_g2(2000, n), R.synth_fr(0, n)
    value = R.product_of_pairings_with_coeffs(a, b, r)
    proof, ch, st = R.SIPP.prove_one_shot(a, b, r, value)
    assert proof.shape == (44, 72)
    assert orc.sipp_verify(a, b, r, value, proof) == 1, "the oracle's verifier rejects the n = 2^22 proof (mem_tier %d)" % st["mem_tier"]
    assert R.SIPP.verify(a, b, r, value, proof)
    bad = proof.copy(); bad[17, 3] ^= 1
    assert orc.sipp_verify(a, b, r, value, bad) == 0
    tier_free, held_free = int(st["mem_tier"]), int(st["device_bytes"])
    R.release_scratch()
    assert R.device_bytes() <= base + (1 << 16)
    # a neighbour takes most of the device: what is left (~60 GB) holds the statement and a cut line buffer, not 57 GB of three-quarter tables
    free, total, ballast = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_void_p()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    taken = max(0, free.value - (60 << 30))
    assert hip.hipMalloc(ctypes.byref(ballast), ctypes.c_size_t(taken)) == 0
    try:
        proof2, ch2, st2 = R.SIPP.prove_one_shot(a, b, r, value)
        assert np.array_equal(proof2, proof) and np.array_equal(ch2, ch), "the proof changed with the memory tier (mem_tier %d)" % st2["mem_tier"]
        assert int(st2["mem_tier"]) > 0, "60 GB cannot hold the three-quarter tables of 2^22 elements: the free memory did not drive the tier"
        print(f"n = 2^22: free device -> mem_tier {tier_free}, {held_free >> 30} GB held; with {taken >> 30} GB taken by a neighbour -> mem_tier {int(st2['mem_tier'])}, {int(st2['device_bytes']) >> 30} GB held")
    finally:
        hip.hipFree(ballast)
        R.release_scratch()
    assert R.device_bytes() <= base + (1 << 16)
