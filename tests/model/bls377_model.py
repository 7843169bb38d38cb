"""Independent big-integer model of the BLS12-377 pairing path (TEST INFRASTRUCTURE ONLY) -- the curve the reference's own SIPP test and
its `scaling-ipp` example run on (sipp/src/lib.rs:229, sipp/examples/scaling-ipp.rs:2,10).

Same construction as bls381_model.py, written from the mathematics:
  * Fp2 = Fp[u]/(u^2 + 5), Fp12 = the flat ring Fp2[w]/(w^6 - u)   (ark-bls12-377 0.4: Fq2 non-residue -5, Fq6 non-residue u),
  * G1: y^2 = x^3 + 1,  G2 on the D-TYPE twist y^2 = x^3 + 1/u, untwist (x', y') -> (x' w^2, y' w^3),
  * Miller loop over the bits of x = +0x8508c00000000001 (no conjugation: x > 0) with the textbook line
        l(P) = yP - yT w^3 - (lambda w)(xP - xT w^2),
  * final exponentiation = ONE pow by (p^6-1)(p^2+1)((x-1)^2 (x+p)(x^2+p^2-1) + 3)  = 3 (p^12-1)/r  (arkworks' generic BLS12 chain),
  * serialisation: the GENERIC ark-ec 0.4 short-Weierstrass layout [ark-mem] -- x then y, little-endian, SWFlags in the two top bits of the
    LAST byte (bit 7: y is the lexicographically larger root, bit 6: infinity), also in the uncompressed form.
The generator coordinates are ark-bls12-377's, recalled and checked here (on the curve, order r); SIPP itself never uses them.
"""
import hashlib
import struct

from bls381_model import FiatShamirRng, chacha20_block  # noqa: F401  (curve independent)

X = 0x8508C00000000001
R = X**4 - X**2 + 1
P = (X - 1) ** 2 * R // 3 + X
assert R == 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
assert P == 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
FINAL_EXP = (P**6 - 1) * (P**2 + 1) * ((X - 1) ** 2 * (X + P) * (X**2 + P**2 - 1) + 3)
assert FINAL_EXP == 3 * ((P**12 - 1) // R)
BETA = -5                       # u^2 = -5
assert pow(BETA % P, (P - 1) // 2, P) == P - 1

F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (0, 1)                     # w^6 = u
def f2(a, b=0): return (a % P, b % P)
def f2add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2mul(a, b): return ((a[0] * b[0] + BETA * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2muls(a, s): return (a[0] * s % P, a[1] * s % P)
def f2inv(a):
    d = pow(a[0] * a[0] - BETA * a[1] * a[1], -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)
B_TWIST = f2inv(XI)             # 1/u = (0, -1/5)
assert B_TWIST == (0, 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906)

F12_ONE = [F2_ONE] + [F2_ZERO] * 5
def f12mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO: continue
        for j in range(6): t[i + j] = f2add(t[i + j], f2mul(a[i], b[j]))
    return [f2add(t[k], f2mul(t[k + 6], XI)) if k < 5 else t[k] for k in range(6)]
def f12pow(a, e):
    res, base = F12_ONE, a
    while e:
        if e & 1: res = f12mul(res, base)
        base = f12mul(base, base); e >>= 1
    return res
def f12_to_tower(a): return [a[0], a[2], a[4], a[1], a[3], a[5]]     # (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2), v = w^2

G1 = (81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
      241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030)
G2 = ((233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
       140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
      (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
       149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491))

class _Fp:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % P); sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P); inv = staticmethod(lambda a: pow(a, -1, P)); muls = staticmethod(lambda a, s: a * s % P)
class _Fp2:
    zero, one = F2_ZERO, F2_ONE
    add, sub, mul, inv, muls = map(staticmethod, (f2add, f2sub, f2mul, f2inv, f2muls))

def ec_add(F, p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if y1 != y2 or y1 == F.zero: return None
        lam = F.mul(F.muls(F.mul(x1, x1), 3), F.inv(F.muls(y1, 2)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    return (x3, F.sub(F.mul(lam, F.sub(x1, x3)), y1))
def ec_neg(F, p): return None if p is None else (p[0], F.sub(F.zero, p[1]))
def ec_mul(F, k, p):
    res, base = None, p
    while k:
        if k & 1: res = ec_add(F, res, base)
        base = ec_add(F, base, base); k >>= 1
    return res
def g1_mul(k, p=G1): return ec_mul(_Fp, k % R, p)
def g2_mul(k, q=G2): return ec_mul(_Fp2, k % R, q)
def g1_add(a, b): return ec_add(_Fp, a, b)
def g2_add(a, b): return ec_add(_Fp2, a, b)
def g1_on_curve(p): return p is None or (p[1] * p[1] - p[0] ** 3 - 1) % P == 0
def g2_on_curve(q): return q is None or f2sub(f2mul(q[1], q[1]), f2add(f2mul(f2mul(q[0], q[0]), q[0]), B_TWIST)) == F2_ZERO
assert g1_on_curve(G1) and g2_on_curve(G2)
def g1_msm(points, scalars):
    acc = None
    for pt, s in zip(points, scalars): acc = g1_add(acc, g1_mul(s, pt))
    return acc
def g2_msm(points, scalars):
    acc = None
    for pt, s in zip(points, scalars): acc = g2_add(acc, g2_mul(s, pt))
    return acc

def _emb2(c, k=0): return [c if i == k else F2_ZERO for i in range(6)]
def _line(T, lam, Pt):
    """line through psi(T) = (xT w^2, yT w^3) with slope lam w, at P:  yP - yT w^3 - lam w (xP - xT w^2)"""
    xP, yP = Pt
    out = [F2_ZERO] * 6
    out[0] = f2(yP)
    out[1] = f2neg(f2muls(lam, xP))
    out[3] = f2sub(f2mul(lam, T[0]), T[1])
    return out
def miller_loop(Pt, Q):
    if Pt is None or Q is None: return F12_ONE
    f, T = F12_ONE, Q
    for bit in bin(X)[3:]:
        lam = f2mul(f2muls(f2mul(T[0], T[0]), 3), f2inv(f2muls(T[1], 2)))
        f = f12mul(f12mul(f, f), _line(T, lam, Pt))
        T = ec_add(_Fp2, T, T)
        if bit == "1":
            lam = f2mul(f2sub(Q[1], T[1]), f2inv(f2sub(Q[0], T[0])))
            f = f12mul(f, _line(T, lam, Pt))
            T = ec_add(_Fp2, T, Q)
    return f                    # x > 0: no conjugation
def final_exponentiation(f): return f12pow(f, FINAL_EXP)
def pairing(Pt, Q): return final_exponentiation(miller_loop(Pt, Q))
def pairing_product(ps, qs):
    f = F12_ONE
    for a, b in zip(ps, qs): f = f12mul(f, miller_loop(a, b))
    return final_exponentiation(f)

# ---- ark-serialize 0.4, generic short-Weierstrass layout [ark-mem]
def ser_fr(s): return (s % R).to_bytes(32, "little")
def ser_fp_le(a): return (a % P).to_bytes(48, "little")
def ser_gt(f): return b"".join(ser_fp_le(c[0]) + ser_fp_le(c[1]) for c in f12_to_tower(f))
def _flag(b, bits): b = bytearray(b); b[-1] |= bits; return bytes(b)
def ser_g1(p):
    if p is None: return bytes(48) + _flag(bytes(48), 0x40)
    neg = p[1] > (P - p[1]) % P                     # `self.y <= -self.y` is YIsPositive
    return ser_fp_le(p[0]) + _flag(ser_fp_le(p[1]), 0x80 if neg else 0)
def _f2_gt(a, b): return (a[1], a[0]) > (b[1], b[0])   # QuadExtField orders by c1, then c0
def ser_g2(q):
    if q is None: return bytes(96) + bytes(48) + _flag(bytes(48), 0x40)
    neg = _f2_gt(q[1], f2neg(q[1]))
    return ser_fp_le(q[0][0]) + ser_fp_le(q[0][1]) + ser_fp_le(q[1][0]) + _flag(ser_fp_le(q[1][1]), 0x80 if neg else 0)
def ser_g1_compressed(p):
    if p is None: return _flag(bytes(48), 0x40)
    return _flag(ser_fp_le(p[0]), 0x80 if p[1] > (P - p[1]) % P else 0)
def ser_g2_compressed(q):
    if q is None: return bytes(48) + _flag(bytes(48), 0x40)
    return ser_fp_le(q[0][0]) + _flag(ser_fp_le(q[0][1]), 0x80 if _f2_gt(q[1], f2neg(q[1])) else 0)
def ser_vec(items, f): return struct.pack("<Q", len(items)) + b"".join(f(i) for i in items)

def sipp_prove(a, b, r, value):
    """sipp/src/lib.rs:42-106 with E = Bls12_377, D = Blake2s."""
    n = len(a)
    assert n == len(b) and n & (n - 1) == 0
    rng = FiatShamirRng(ser_vec(a, ser_g1) + ser_vec(b, ser_g2) + ser_vec(r, ser_fr) + ser_gt(value))
    a = [g1_mul(ri, ai) for ai, ri in zip(a, r)]; b = list(b)
    proof, challenges = [], []
    while n != 1:
        n //= 2
        a_l, a_r, b_l, b_r = a[:n], a[n:], b[:n], b[n:]
        z_l, z_r = pairing_product(a_r, b_l), pairing_product(a_l, b_r)
        proof.append((z_l, z_r))
        rng.absorb(ser_gt(z_l) + ser_gt(z_r))
        x = rng.next_u128() % R
        challenges.append(x)
        x_inv = pow(x, -1, R)
        a = [g1_add(g1_mul(x, ar), al) for al, ar in zip(a_l, a_r)]
        b = [g2_add(g2_mul(x_inv, br), bl) for bl, br in zip(b_l, b_r)]
    return proof, challenges
