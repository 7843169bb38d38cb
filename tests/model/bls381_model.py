"""Independent big-integer model of the BLS12-381 pairing path (TEST INFRASTRUCTURE ONLY).

Written from the mathematics, not from the C oracle nor from the HIP code:
  * Fp12 is the flat polynomial ring Fp2[w]/(w^6 - (1+u))  (no Fp6 tower),
  * the Miller loop uses affine twist arithmetic and the textbook untwisted line
    l(P) = yP - yT/w^3 - (lambda/w)(xP - xT/w^2),
  * the final exponentiation is ONE big `pow` by the integer
        (p^6-1)(p^2+1) * ((x-1)^2 (x+p)(x^2+p^2-1) + 3)
    which is the exponent ark-ec 0.4's `Bls12::final_exponentiation` realises
    (the cube of the textbook optimal-ate pairing), see SURVEY.md section 8(c) item 4.
It exists to cross-check oracle/ (C) and to generate tests/golden/*.json.

Reference call sites this models: inner_products/src/lib.rs:77-116 (pairing product),
inner_products/src/lib.rs:128-141 (MSM), sipp/src/lib.rs:42-106 (SIPP prover),
sipp/src/rng.rs:47-72 (FiatShamirRng).
"""
import hashlib
import struct

X_ABS = 0xD201000000010000
X = -X_ABS
R = X**4 - X**2 + 1
P = (X - 1) ** 2 * R // 3 + X
FINAL_EXP = (P**6 - 1) * (P**2 + 1) * ((X - 1) ** 2 * (X + P) * (X**2 + P**2 - 1) + 3)
assert FINAL_EXP == 3 * ((P**12 - 1) // R)

# ------------------------------------------------------------------ Fp2 = Fp[u]/(u^2+1)
def f2(a, b=0):
    return (a % P, b % P)

F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (1, 1)

def f2add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2muls(a, s): return (a[0] * s % P, a[1] * s % P)
def f2inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)

# ------------------------------------------------------------------ Fp12 = Fp2[w]/(w^6 - XI), list of 6 Fp2
F12_ONE = [F2_ONE] + [F2_ZERO] * 5

def f12mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO:
            continue
        for j in range(6):
            t[i + j] = f2add(t[i + j], f2mul(a[i], b[j]))
    return [f2add(t[k], f2mul(t[k + 6], XI)) if k < 5 else t[k] for k in range(6)]

def f12pow(a, e):
    res, base = F12_ONE, a
    while e:
        if e & 1:
            res = f12mul(res, base)
        base = f12mul(base, base)
        e >>= 1
    return res

def f12conj(a):  # p^6-Frobenius: w -> -w
    return [a[k] if k % 2 == 0 else f2neg(a[k]) for k in range(6)]

def f12_to_tower(a):
    """flat sum g_k w^k  ->  arkworks tower order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2), v = w^2."""
    return [a[0], a[2], a[4], a[1], a[3], a[5]]

# ------------------------------------------------------------------ curves (affine, None = infinity)
G1 = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
      0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
G2 = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
       0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
      (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
       0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))

class _Fp:  # field-op bundle so one set of curve formulas serves G1 and G2
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    inv = staticmethod(lambda a: pow(a, -1, P))
    muls = staticmethod(lambda a, s: a * s % P)

class _Fp2:
    zero, one = F2_ZERO, F2_ONE
    add, sub, mul, inv, muls = map(staticmethod, (f2add, f2sub, f2mul, f2inv, f2muls))

def ec_add(F, p1, p2):
    if p1 is None: return p2
    if p2 is None: return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if y1 != y2 or y1 == F.zero:
            return None
        lam = F.mul(F.muls(F.mul(x1, x1), 3), F.inv(F.muls(y1, 2)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    return (x3, F.sub(F.mul(lam, F.sub(x1, x3)), y1))

def ec_neg(F, p): return None if p is None else (p[0], F.sub(F.zero, p[1]))

def ec_mul(F, k, p):
    if k < 0: return ec_mul(F, -k, ec_neg(F, p))
    res, base = None, p
    while k:
        if k & 1: res = ec_add(F, res, base)
        base = ec_add(F, base, base)
        k >>= 1
    return res

def g1_mul(k, p=G1): return ec_mul(_Fp, k % R, p)
def g2_mul(k, q=G2): return ec_mul(_Fp2, k % R, q)
def g1_add(a, b): return ec_add(_Fp, a, b)
def g2_add(a, b): return ec_add(_Fp2, a, b)
def g1_on_curve(p): return p is None or (p[1] * p[1] - p[0] ** 3 - 4) % P == 0
def g2_on_curve(q):
    if q is None: return True
    return f2sub(f2mul(q[1], q[1]), f2add(f2mul(f2mul(q[0], q[0]), q[0]), f2muls(XI, 4))) == F2_ZERO

def g1_msm(points, scalars):
    acc = None
    for pt, s in zip(points, scalars): acc = g1_add(acc, g1_mul(s, pt))
    return acc
def g2_msm(points, scalars):
    acc = None
    for pt, s in zip(points, scalars): acc = g2_add(acc, g2_mul(s, pt))
    return acc

# ------------------------------------------------------------------ pairing
_W_INV = None
def _w_pows():
    """w^-1, w^-2, w^-3 as flat Fp12 elements: w^-1 = w^5 / XI."""
    global _W_INV
    if _W_INV is None:
        xi_inv = f2inv(XI)
        w1 = [F2_ZERO] * 5 + [xi_inv]
        w2 = f12mul(w1, w1)
        _W_INV = (w1, w2, f12mul(w2, w1))
    return _W_INV

def _emb2(c, k=0):  # Fp2 element times w^k (k in 0..5)
    return [c if i == k else F2_ZERO for i in range(6)]

def _line(T, lam, Pt):
    """Untwisted line through psi(T) with twist-slope lam, evaluated at P in E(Fp)."""
    w1, w2, w3 = _w_pows()
    xP, yP = Pt
    xT = f12mul(_emb2(T[0]), w2)      # xT / w^2
    yT = f12mul(_emb2(T[1]), w3)      # yT / w^3
    lamE = f12mul(_emb2(lam), w1)     # lambda / w
    dx = [f2sub(f2(xP) if k == 0 else F2_ZERO, xT[k]) for k in range(6)]
    t = f12mul(lamE, dx)
    return [f2sub(f2sub(f2(yP) if k == 0 else F2_ZERO, yT[k]), t[k]) for k in range(6)]

def miller_loop(Pt, Q):
    """f_{|x|,Q}(P) conjugated (x < 0); infinity on either side contributes 1."""
    if Pt is None or Q is None:
        return F12_ONE
    f, T = F12_ONE, Q
    for bit in bin(X_ABS)[3:]:
        lam = f2mul(f2muls(f2mul(T[0], T[0]), 3), f2inv(f2muls(T[1], 2)))
        f = f12mul(f12mul(f, f), _line(T, lam, Pt))
        T = ec_add(_Fp2, T, T)
        if bit == "1":
            lam = f2mul(f2sub(Q[1], T[1]), f2inv(f2sub(Q[0], T[0])))
            f = f12mul(f, _line(T, lam, Pt))
            T = ec_add(_Fp2, T, Q)
    return f12conj(f)

def final_exponentiation(f): return f12pow(f, FINAL_EXP)
def pairing(Pt, Q): return final_exponentiation(miller_loop(Pt, Q))
def pairing_product(ps, qs):
    f = F12_ONE
    for a, b in zip(ps, qs): f = f12mul(f, miller_loop(a, b))
    return final_exponentiation(f)

# ------------------------------------------------------------------ ark-serialize 0.4 `serialize_uncompressed`
def ser_fr(s): return (s % R).to_bytes(32, "little")
def ser_fp_le(a): return (a % P).to_bytes(48, "little")
def ser_gt(f):
    return b"".join(ser_fp_le(c[0]) + ser_fp_le(c[1]) for c in f12_to_tower(f))
def ser_g1(p):  # ark-bls12-381 0.4: zcash layout, big-endian, flags in the top bits of byte 0
    if p is None: return bytes([0x40]) + bytes(95)
    return p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")
def ser_g2(q):
    if q is None: return bytes([0x40]) + bytes(191)
    return q[0][1].to_bytes(48, "big") + q[0][0].to_bytes(48, "big") + q[1][1].to_bytes(48, "big") + q[1][0].to_bytes(48, "big")
def ser_vec(items, f): return struct.pack("<Q", len(items)) + b"".join(f(i) for i in items)

# ------------------------------------------------------------------ ChaCha20 (djb: 64-bit counter, 64-bit stream id) as rand_chacha 0.3 ChaCha20Rng
def _rotl(v, c): return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))
def chacha20_block(key32, counter):
    k = struct.unpack("<8I", key32)
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574, *k, counter & 0xFFFFFFFF, counter >> 32, 0, 0]
    w = list(s)
    def qr(a, b, c, d):
        w[a] = (w[a] + w[b]) & 0xFFFFFFFF; w[d] = _rotl(w[d] ^ w[a], 16)
        w[c] = (w[c] + w[d]) & 0xFFFFFFFF; w[b] = _rotl(w[b] ^ w[c], 12)
        w[a] = (w[a] + w[b]) & 0xFFFFFFFF; w[d] = _rotl(w[d] ^ w[a], 8)
        w[c] = (w[c] + w[d]) & 0xFFFFFFFF; w[b] = _rotl(w[b] ^ w[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(w[i] + s[i]) & 0xFFFFFFFF for i in range(16)])

class FiatShamirRng:
    """sipp/src/rng.rs:47-72 with D = Blake2s."""
    def __init__(self, seed_bytes):
        self.seed = hashlib.blake2s(seed_bytes).digest()
        self.pos = 0
    def absorb(self, new_bytes):
        self.seed = hashlib.blake2s(new_bytes + self.seed).digest()   # new material FIRST (rng.rs:68-70)
        self.pos = 0
    def next_bytes(self, n):
        out = b""
        while len(out) < n:
            blk, off = divmod(self.pos, 64)
            chunk = chacha20_block(self.seed, blk)[off:off + (n - len(out))]
            out += chunk; self.pos += len(chunk)
        return out
    def next_u128(self):  # rand 0.8 Standard: low u64 first, each u64 = two LE u32 words
        return int.from_bytes(self.next_bytes(16), "little")

# ------------------------------------------------------------------ SIPP prover (sipp/src/lib.rs:42-106)
def sipp_prove(a, b, r, value):
    n = len(a)
    assert n == len(b) and n & (n - 1) == 0
    seed = ser_vec(a, ser_g1) + ser_vec(b, ser_g2) + ser_vec(r, ser_fr) + ser_gt(value)
    rng = FiatShamirRng(seed)
    a = [g1_mul(ri, ai) for ai, ri in zip(a, r)]
    b = list(b)
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
