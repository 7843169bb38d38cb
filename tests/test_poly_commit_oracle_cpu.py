"""CPU: the oracle-side restatement of the polynomial-commitment applications (tests/model/poly_commit_oracle.py) is self-consistent --
the reference's own tests (poly_commit/mod.rs:405-472, transparent.rs:346-413) at small degrees: commit, open at a random point, verify;
a wrong evaluation and a wrong point are rejected."""
import os, random, sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "model"))
import orclib as o
import poly_commit_oracle as PC


def _poly(rng, n): return [rng.randrange(o.R) for _ in range(n)]


@pytest.mark.parametrize("degree", [0, 1, 5])
def test_kzg_round_trip(degree):
    rng = random.Random(degree)
    powers, v = PC.kzg_setup(rng.randrange(1, o.R), rng.randrange(1, o.R), 7)
    p = _poly(rng, degree + 1); z = rng.randrange(o.R)
    com = PC.kzg_commit(powers, p); proof = PC.kzg_open(powers, p, z)
    assert PC.kzg_verify(v, com, z, PC.horner(p, z), proof)
    assert not PC.kzg_verify(v, com, z, (PC.horner(p, z) + 1) % o.R, proof)


def test_bivariate_poly_commit_small():
    """mod.rs:405-443 at x_degree = 3, y_degree = 3"""
    rng = random.Random(7)
    s = PC.bi_setup(rng.randrange(1, o.R), rng.randrange(1, o.R), 3, 3)
    ys = [_poly(rng, 4) for _ in range(4)]
    com, coms = PC.bi_commit(s, ys)
    point = (rng.randrange(o.R), rng.randrange(o.R))
    proof = PC.bi_open(s, ys, coms, point); ev = PC.bi_evaluate(ys, point)
    assert PC.bi_verify(s["v"], com, point, ev, proof)
    assert not PC.bi_verify(s["v"], com, point, (ev + 1) % o.R, proof)
    assert not PC.bi_verify(s["v"], com, (point[0], (point[1] + 1) % o.R), ev, proof)


def test_transparent_bivariate_poly_commit_small():
    """transparent.rs:346-379 at x_degree = 1, y_degree = 3"""
    rng = random.Random(8)
    ck1 = o.to_jac_g1(o.gen_g1(700, 4)); ck2 = o.to_jac_g2(o.gen_g2(900, 2))
    ys = [_poly(rng, 4) for _ in range(2)]
    com, coms = PC.tr_commit(ck1, ck2, ys)
    point = (rng.randrange(o.R), rng.randrange(o.R))
    proof = PC.tr_open(ck1, ck2, ys, coms, point); ev = PC.bi_evaluate(ys, point)
    assert PC.tr_verify(ck1, ck2, com, point, ev, proof)
    assert not PC.tr_verify(ck1, ck2, com, point, (ev + 1) % o.R, proof)
