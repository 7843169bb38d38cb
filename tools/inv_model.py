"""Limb-level model (14 limbs of 28 bits, Python integers) of the device inversion ripp_amd/csrc/bls12_381/fp_inv.hpp::fp_inv_bingcd:
binary GCD with K = 30 inner steps on 64-bit approximations (low 30 bits + the 34 bits below the top bit of a | b), 26 outer iterations,
exact division by 2^30 of (a, b) and modular division of (u, v) per outer iteration (Pornin, eprint 2020/972, Alg. 2).  Every step asserts
the invariants the device code relies on (factor bounds, exact divisibility, |u|, |v| < 64 p, a = 0 and b = 1 at the end)."""
import random
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
W, NL, MASK = 28, 14, (1 << 28) - 1
K = 30
ITER = (2 * 381 - 1 + K - 1) // K
MINV = (-pow(P, -1, 1 << K)) % (1 << K)
PL = [(P >> (W * i)) & MASK for i in range(NL)]

def to_limbs(x): return [(x >> (W * i)) & MASK for i in range(NL)]
def val(l): return sum(v << (W * i) for i, v in enumerate(l))       # top limb may be negative

def carry(col):           # signed carry pass: limbs 0..12 in [0, 2^28), top limb signed
    r, c = [], 0
    for i in range(NL - 1):
        t = col[i] + c; r.append(t & MASK); c = t >> W
    r.append(col[NL - 1] + c)
    return r

def shr30(l):             # exact division by 2^30 of a value whose low 30 bits are zero (limbs normalised, top signed)
    assert l[0] == 0 and (l[1] & 3) == 0
    r = []
    for i in range(NL):
        lo = l[i + 1] >> 2 if i + 1 < NL else (l[NL - 1] >> 30)          # arithmetic for the top limb
        hi = (l[i + 2] << 26) if i + 2 < NL else ((l[NL - 1] >> 2) >> 28 << 26 if False else 0)
        r.append(lo | hi)
    # simpler and exact: do it on the integer, then re-split keeping the sign in the top limb
    v = val(l) >> 30
    out = [(v >> (W * i)) & MASK for i in range(NL - 1)] + [v >> (W * (NL - 1))]
    return out

def top34_low30(a, b):
    # a, b non-negative, limbs normalised.  n = bit length of (a | b)
    w = NL - 1
    while w > 0 and (a[w] | b[w]) == 0: w -= 1
    if w <= 1:            # n <= 56: exact
        return a[0] | (a[1] << W), b[0] | (b[1] << W)
    # three limbs w, w-1, w-2 (84 bits); top limb nonzero in a|b
    ta = (a[w] << 56) | (a[w - 1] << 28) | a[w - 2]; tb = (b[w] << 56) | (b[w - 1] << 28) | b[w - 2]
    bl = (a[w] | b[w]).bit_length()               # 1..28
    sh = 56 + bl - 34                             # keep the top 34 bits of the (56 + bl)-bit window
    if w == 2 and sh < 0: sh = 0
    ha, hb = ta >> sh, tb >> sh
    n = W * (w - 2) + 56 + bl
    if n <= 64: return val(a) & ((1 << 64) - 1), val(b) & ((1 << 64) - 1)
    lo30 = (1 << 30) - 1
    la = (a[0] | (a[1] << W)) & lo30; lb = (b[0] | (b[1] << W)) & lo30
    return la | (ha << 30), lb | (hb << 30)

def inv(y):
    a, b = to_limbs(y), to_limbs(P)
    u, v = to_limbs(1), to_limbs(0)
    for it in range(ITER):
        a_, b_ = top34_low30(a, b)
        assert a_ < (1 << 64) and b_ < (1 << 64)
        f0, g0, f1, g1 = 1, 0, 0, 1
        for i in range(K):
            if a_ & 1:
                if a_ < b_: a_, b_ = b_, a_; f0, f1 = f1, f0; g0, g1 = g1, g0
                a_ -= b_; f0 -= f1; g0 -= g1
            a_ >>= 1; f1 *= 2; g1 *= 2
        assert max(abs(f0), abs(g0), abs(f1), abs(g1)) <= (1 << 30)
        na = shr30(carry([f0 * a[i] + g0 * b[i] for i in range(NL)])); nb = shr30(carry([f1 * a[i] + g1 * b[i] for i in range(NL)]))
        if na[NL - 1] < 0: na = carry([-x for x in na]); f0, g0 = -f0, -g0
        if nb[NL - 1] < 0: nb = carry([-x for x in nb]); f1, g1 = -f1, -g1
        a, b = na, nb
        for (ff, gg, which) in ((f0, g0, 0), (f1, g1, 1)):
            col = [ff * u[i] + gg * v[i] for i in range(NL)]
            low = (col[0] + (col[1] << W)) & ((1 << K) - 1)
            k = (low * MINV) & ((1 << K) - 1)
            col = [col[i] + k * PL[i] for i in range(NL)]
            r = shr30(carry(col))
            if which == 0: nu = r
            else: nv = r
        u, v = nu, nv
        assert abs(val(u)) < 64 * P and abs(val(v)) < 64 * P
    assert val(a) == 0 and val(b) == 1
    return val(v) % P

def self_test(samples=3000):
    rnd = random.Random(2)
    for y in [1, 2, 3, P - 1, P - 2, (P + 1) // 2, 1 << 380, (1 << 381) - 1 - (1 << 300), 1 << 56, (1 << 57) + 1, 1 << 84, 5 << 100] + [rnd.randrange(1, P) for _ in range(samples)] + [rnd.randrange(1, 1 << rnd.randrange(1, 381)) for _ in range(samples)]:
        assert inv(y) * y % P == 1, hex(y)
    return ITER


if __name__ == "__main__":
    print("ok", self_test(), hex(MINV))
