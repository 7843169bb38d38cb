"""CPU restatement of the polynomial-commitment applications (ip_proofs/src/applications/poly_commit/mod.rs and transparent.rs)
over the C oracle -- test infrastructure, the checker for ripp_amd/poly_commit.  Written independently of that package: polynomials are
lists of integers, every group operation goes through tests/orclib.py, the argument systems are the oracle's own
(orc_tipa_ssm_prove / verify for the pairing-based scheme, tests/model/gipa_generic_oracle.py for the transparent one)."""
import numpy as np
import orclib as o
import gipa_generic_oracle as M

R = o.R
SECOND_TIER = ("MEXP1", "AFGHO1", "SSM", "G1")       # transparent.rs:28-33
FIRST_TIER = ("SCAL", "PED1", "SSM", "FR")           # transparent.rs:43-48


def horner(coeffs, z):
    acc = 0
    for c in coeffs[::-1]:
        acc = (acc * z + c) % R
    return acc


def pad(c, n): return list(c) + [0] * (n - len(c))
def g1_neg_mul(p_aff_or_jac, k): return M.scale("G1", p_aff_or_jac, (-k) % R)


# ---- KZG (mod.rs:50-119)
def kzg_setup(alpha, beta, degree):
    fa = o.fr_array([alpha])[0]
    powers = o.normalize_g1(o.srs_powers_g1(fa, degree + 1))
    g, h = o.to_jac_g1(o.g1_generator())[0], o.to_jac_g2(o.g2_generator())[0]
    v_srs = dict(g=g, h=h, g_beta=M.scale("G1", g, beta), h_alpha=M.scale("G2", h, alpha))
    return powers, v_srs


def kzg_commit(powers, poly):
    return o.msm_g1_a(np.ascontiguousarray(powers), o.fr_array(pad(poly, len(powers))))


def kzg_open(powers, poly, z):
    """quotient of p(X) by (X - z) by long division from the top coefficient (mod.rs:96-104)"""
    rem = list(poly); q = [0] * max(len(poly) - 1, 0)
    for i in range(len(poly) - 1, 0, -1):
        q[i - 1] = rem[i] % R
        rem[i - 1] = (rem[i - 1] + rem[i] * z) % R
    return o.msm_g1_a(np.ascontiguousarray(powers), o.fr_array(pad(q, len(powers))))


def kzg_verify(v, com, z, ev, proof):
    lhs_p = M.plus("G1", com, M.scale("G1", v["g"], (-ev) % R))
    rhs_q = M.plus("G2", v["h_alpha"], M.scale("G2", v["h"], (-z) % R))
    lhs = o.pairing_product_j(lhs_p[None], v["h"][None])[1]
    rhs = o.pairing_product_j(np.ascontiguousarray(proof)[None], rhs_q[None])[1]
    return bool(np.array_equal(lhs, rhs))


# ---- pairing-based bivariate commitment (mod.rs:142-296)
def bi_setup(alpha, beta, x_degree, y_degree):
    fb = o.fr_array([beta])[0]
    kzg_srs, v = kzg_setup(alpha, beta, y_degree)
    hbp = o.srs_powers_g2(fb, 2 * x_degree + 1)
    return dict(h_beta_powers=hbp, ck=np.ascontiguousarray(hbp[::2]), kzg=kzg_srs, v=v)


def rows_of(y_polys, n_rows, n_cols):
    return [pad(p, n_cols) for p in y_polys] + [[0] * n_cols for _ in range(n_rows - len(y_polys))]


def bi_commit(s, y_polys):
    rows = rows_of(y_polys, len(s["ck"]), len(s["kzg"]))
    coms = np.stack([kzg_commit(s["kzg"], r) for r in rows])
    return o.pairing_product_j(coms, s["ck"])[1], coms


def partial_eval(rows, x, n_cols):
    out = [0] * n_cols; xp = 1
    for r in rows:
        for j in range(n_cols):
            out[j] = (out[j] + xp * r[j]) % R
        xp = xp * x % R
    return out


def bi_open(s, y_polys, coms, point):
    x, y = point; n = len(s["ck"])
    rows = rows_of(y_polys, n, len(s["kzg"]))
    ye = partial_eval(rows, x, len(s["kzg"]))
    y_eval_comm = kzg_commit(s["kzg"], ye)
    rc, ip = o.tipa_ssm_prove(s["h_beta_powers"], np.ascontiguousarray(coms), o.fr_array([pow(x, i, R) for i in range(n)]), s["ck"])
    assert rc == 0
    return dict(ip_proof=ip, y_eval_comm=y_eval_comm, kzg_proof=kzg_open(s["kzg"], ye, y))


def bi_verify(v, com, point, ev, proof):
    x, y = point
    ok_ip = o.tipa_ssm_verify(v["g"], v["h"], v["g_beta"], np.ascontiguousarray(com), np.ascontiguousarray(proof["y_eval_comm"]), o.fr_array([x])[0], proof["ip_proof"]) == 1
    return ok_ip and kzg_verify(v, proof["y_eval_comm"], y, ev, proof["kzg_proof"])


def bi_evaluate(y_polys, point):
    x, y = point
    return sum(pow(x, i, R) * horner(p, y) for i, p in enumerate(y_polys)) % R


def split(poly, x_degree, y_degree):
    flat = pad(poly, (x_degree + 1) * (y_degree + 1))
    return [flat[i * (y_degree + 1):(i + 1) * (y_degree + 1)] for i in range(x_degree + 1)]


# ---- transparent scheme (transparent.rs:86-225)
def tr_commit(ck1, ck2, y_polys):
    rows = rows_of(y_polys, len(ck2), len(ck1))
    coms = np.stack([o.msm_g1_j(np.ascontiguousarray(ck1), o.fr_array(r))[1] for r in rows])
    return o.pairing_product_j(coms, np.ascontiguousarray(ck2))[1], coms


def tr_open(ck1, ck2, y_polys, coms, point):
    x, y = point
    rows = rows_of(y_polys, len(ck2), len(ck1))
    ye = partial_eval(rows, x, len(ck1))
    y_eval_comm = o.msm_g1_j(np.ascontiguousarray(ck1), o.fr_array(ye))[1]
    second = M.prove(SECOND_TIER, np.ascontiguousarray(coms), [pow(x, i, R) for i in range(len(ck2))], np.ascontiguousarray(ck2), [None] * len(ck2))
    first = M.prove(FIRST_TIER, ye, [pow(y, i, R) for i in range(len(ck1))], np.ascontiguousarray(ck1), [None] * len(ck1))
    return dict(second=second, y_eval_comm=y_eval_comm, first=first)


def tr_verify(ck1, ck2, com, point, ev, proof):
    x, y = point
    s_steps, _, s_base, _ = proof["second"]; f_steps, _, f_base, _ = proof["first"]
    ok2 = M.verify(SECOND_TIER, np.ascontiguousarray(ck2), [None] * len(ck2), [com, 0, proof["y_eval_comm"]], s_steps, s_base, scalar_b=x)
    ok1 = M.verify(FIRST_TIER, np.ascontiguousarray(ck1), [None] * len(ck1), [proof["y_eval_comm"], 0, ev % R], f_steps, f_base, scalar_b=y)
    return ok2 and ok1
