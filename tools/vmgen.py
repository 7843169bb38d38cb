#!/usr/bin/env python3
"""Static scheduler for the lane-parallel field VM (ripp_amd/csrc/vm.hpp).

Idea: G lanes of a wave cooperate on ONE element (a (P,Q) pair, a fold term, an Fp12 accumulator).  All Fp values
of the element live in a per-element LDS workspace ("slots" of 48 B); a program is a sequence of LAYERS, and in one
layer every lane executes one operation of the same kind:

    MUL   slot[dst] = (+-slot[a0] +-slot[a1]) * (+-slot[a2] +-slot[a3])        (Montgomery product in Fp)
    LIN   slot[dst] = (+-slot[a0] +-slot[a1] +-slot[a2] +-slot[a3]) [/ 2]

Formulas are written once in a tiny DSL over lazy linear forms (so Karatsuba sums, (1+u) twists, negations and small
multiples cost nothing until they must be materialised); this file list-schedules the resulting DAG into layers for a
given group size G, allocates workspace slots by liveness, EVALUATES the schedule with Python integers against the
plain formulas (so a wrong schedule never reaches the GPU), and emits the tables as a C++ header.

Run:  python tools/vmgen.py   ->  ripp_amd/csrc/vm_programs.inc
"""
import os
import random
import sys

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
MUL, LIN = 0, 1
ZERO_SLOT = 0          # workspace slot 0 always holds 0; slot 1 is a write-only dump for idle lanes
DUMP_SLOT = 1
FIRST_FREE = 2


# ------------------------------------------------------------------------------------------------ DSL
class Val:
    """A materialised Fp value (an input or the result of one VM op)."""
    _n = 0

    def __init__(self, prog, kind, args=None, name=None, half=False, sh=0):
        self.prog, self.kind, self.args, self.name, self.half, self.sh = prog, kind, args, name, half, sh
        self.id = Val._n; Val._n += 1
        self.slot = None

    def __repr__(self):
        return f"v{self.id}" + (f"({self.name})" if self.name else "")


class Lin:
    """Lazy linear form sum c_i * Val_i with small integer coefficients."""

    def __init__(self, terms=None):
        self.t = {k: v for k, v in (terms or {}).items() if v != 0}

    @staticmethod
    def of(v): return v if isinstance(v, Lin) else Lin({v: 1})
    def __add__(self, o):
        o = Lin.of(o); t = dict(self.t)
        for k, v in o.t.items(): t[k] = t.get(k, 0) + v
        return Lin(t)
    def __sub__(self, o): return self + (Lin.of(o) * -1)
    def __neg__(self): return self * -1
    def __mul__(self, c): return Lin({k: v * c for k, v in self.t.items()})
    def nterms(self): return sum(abs(c) for c in self.t.values())
    def simple(self):  # encodable as a MUL operand: at most two +-1 terms
        return self.nterms() <= 2
    def expand(self):  # list of (sign, Val) with |coeff| repeats
        out = []
        for k, c in sorted(self.t.items(), key=lambda kv: kv[0].id):
            out += [(1 if c > 0 else -1, k)] * abs(c)
        return out


class Prog:
    def __init__(self, name):
        self.name = name; self.inputs = {}; self.outputs = []; self.nodes = []

    def inp(self, name):
        v = Val(self, "in", name=name); self.inputs[name] = v; return Lin.of(v)

    def materialise(self, L, half=False):
        """Return a Lin of ONE Val equal to L (or L/2), emitting LIN ops (<= 4 signed terms each) as needed."""
        L = Lin.of(L)
        if not half and L.nterms() == 1 and list(L.t.values())[0] == 1:
            return L
        # factor the gcd of the coefficients: g = odd * 2^k.  L/g is summed first (few terms), the odd factor (3) is a
        # second 3-term op and the power of two rides along as the op's shift -- keeps LIN chains short.
        from math import gcd
        g = 0
        for c in L.t.values(): g = gcd(g, abs(c))
        k = 0
        while g % 2 == 0 and k < 3: g //= 2; k += 1
        odd = g if g in (1, 3) else 1
        scale = odd << k
        L = Lin({kk: c // scale for kk, c in L.t.items()})
        if half and k > 0: k -= 1; half = False
        terms = L.expand()
        while len(terms) > 4:  # fold the first four terms into one value
            v = Val(self, LIN, args=terms[:4]); self.nodes.append(v); terms = [(1, v)] + terms[4:]
        if odd == 3 and len(terms) > 1:
            v = Val(self, LIN, args=terms); self.nodes.append(v); terms = [(1, v)] * 3
        elif odd == 3:
            terms = terms * 3
        v = Val(self, LIN, args=terms, half=half, sh=k); self.nodes.append(v)
        return Lin.of(v)

    def operand(self, L):
        L = Lin.of(L)
        return L if L.simple() else self.materialise(L)

    def mul(self, A, B):
        A, B = self.operand(A), self.operand(B)
        v = Val(self, MUL, args=(A.expand(), B.expand())); self.nodes.append(v); return Lin.of(v)

    def half(self, L): return self.materialise(L, half=True)

    def out(self, name, L, into=None):
        """Declare an output; `into` = name of the input whose slot it must end up in (loop-carried state)."""
        L = self.materialise(L) if not (Lin.of(L).nterms() == 1 and list(Lin.of(L).t.values())[0] == 1 and list(Lin.of(L).t.keys())[0].kind != "in") else Lin.of(L)
        self.outputs.append((name, list(L.t.keys())[0], into))


class F2:
    """Fp2 = Fp[u]/(u^2+1) over lazy linear forms."""

    def __init__(self, p, c0, c1): self.p, self.c0, self.c1 = p, Lin.of(c0), Lin.of(c1)
    def __add__(self, o): return F2(self.p, self.c0 + o.c0, self.c1 + o.c1)
    def __sub__(self, o): return F2(self.p, self.c0 - o.c0, self.c1 - o.c1)
    def __neg__(self): return F2(self.p, -self.c0, -self.c1)
    def scale(self, c): return F2(self.p, self.c0 * c, self.c1 * c)
    def mul_xi(self): return F2(self.p, self.c0 - self.c1, self.c0 + self.c1)           # * (1 + u)
    def mul(self, o):                                                                    # Karatsuba, 3 products
        t0, t1 = self.p.mul(self.c0, o.c0), self.p.mul(self.c1, o.c1)
        m = self.p.mul(self.c0 + self.c1, o.c0 + o.c1)
        return F2(self.p, t0 - t1, m - t0 - t1)
    def sqr(self):                                                                       # (a0+a1)(a0-a1), 2 a0 a1
        m = self.p.mul(self.c0, self.c1)
        return F2(self.p, self.p.mul(self.c0 + self.c1, self.c0 - self.c1), m * 2)
    def mul_fp(self, s): return F2(self.p, self.p.mul(self.c0, s), self.p.mul(self.c1, s))
    def half(self): return F2(self.p, self.p.half(self.c0), self.p.half(self.c1))
    def mat(self): return F2(self.p, self.p.materialise(self.c0), self.p.materialise(self.c1))


class F1:
    """Fp itself behind the F2 interface (for the G1 programs)."""

    def __init__(self, p, c0): self.p, self.c0 = p, Lin.of(c0)
    def __add__(self, o): return F1(self.p, self.c0 + o.c0)
    def __sub__(self, o): return F1(self.p, self.c0 - o.c0)
    def __neg__(self): return F1(self.p, -self.c0)
    def scale(self, c): return F1(self.p, self.c0 * c)
    def mul_xi(self): return self                               # G1: b = 4, no twist factor
    def mul(self, o): return F1(self.p, self.p.mul(self.c0, o.c0))
    def sqr(self): return F1(self.p, self.p.mul(self.c0, self.c0))
    def half(self): return F1(self.p, self.p.half(self.c0))
    def mat(self): return F1(self.p, self.p.materialise(self.c0))


# ------------------------------------------------------------------------------------------------ scheduling
def schedule(prog, G):
    """List-schedule prog.nodes into homogeneous layers of <= G ops.  Returns list of (kind, [Val])."""
    nodes = prog.nodes
    deps = {}
    for v in nodes:
        srcs = [t[1] for t in v.args] if v.kind == LIN else [t[1] for t in v.args[0] + v.args[1]]
        deps[v] = {s for s in srcs if s.kind != "in"}
    # critical-path priority (longest path to a sink, MUL weighted 4, LIN 1)
    users = {v: [] for v in nodes}
    for v in nodes:
        for s in deps[v]: users[s].append(v)
    prio = {}
    for v in reversed(nodes):
        prio[v] = (4 if v.kind == MUL else 1) + max([prio[u] for u in users[v]], default=0)
    done, layers, remaining = set(), [], list(nodes)
    while remaining:
        # (1) every LIN chain that is ready, level by level, each level packed into ceil(n/G) layers
        while True:
            rl = [v for v in remaining if v.kind == LIN and deps[v] <= done]
            if not rl: break
            rl.sort(key=lambda v: -prio[v])
            for i in range(0, len(rl), G): layers.append((LIN, rl[i:i + G]))
            done |= set(rl); remaining = [v for v in remaining if v not in done]
        # (2) every product that is ready now
        rm = [v for v in remaining if v.kind == MUL and deps[v] <= done]
        if rm:
            rm.sort(key=lambda v: -prio[v])
            for i in range(0, len(rm), G): layers.append((MUL, rm[i:i + G]))
            done |= set(rm); remaining = [v for v in remaining if v not in done]
    return layers


def allocate(prog, layers, nslots_hint=None):
    """Liveness-based slot allocation.  Inputs get fixed slots FIRST_FREE.. in declaration order."""
    slot = {}
    nxt = FIRST_FREE
    for name, v in prog.inputs.items():
        slot[v] = nxt; nxt += 1
    last_use = {}
    for li, (_, ops) in enumerate(layers):
        for v in ops:
            srcs = [t[1] for t in v.args] if v.kind == LIN else [t[1] for t in v.args[0] + v.args[1]]
            for s in srcs: last_use[s] = li
    outvals = {v for _, v, _ in prog.outputs}
    pinned_into = {v: prog.inputs[into] for _, v, into in prog.outputs if into}
    free = []
    live_inputs = set(prog.inputs.values())
    for li, (_, ops) in enumerate(layers):
        # values (incl. inputs not loop-carried... inputs are never freed: they are the caller's) whose last use was an earlier layer
        for v, s in list(slot.items()):
            if v in live_inputs or v in outvals: continue
            if last_use.get(v, -1) < li and s is not None and s not in free and s >= FIRST_FREE and v.slot_released is False:
                free.append(s); v.slot_released = True
        for v in ops:
            # reads of this layer happen before its writes: a value last used IN this layer may donate its slot
            cand = None
            if v in pinned_into:
                tgt = pinned_into[v]
                if last_use.get(tgt, -1) <= li: cand = slot[tgt]            # overwrite the loop-carried input in place
            if cand is None:
                if free: cand = free.pop(0)
                else: cand = nxt; nxt += 1
            slot[v] = cand; v.slot_released = False
    for v in prog.inputs.values(): v.slot_released = False
    fixups = []   # outputs that could not be written in place: copy at the end
    for name, v, into in prog.outputs:
        if into and slot[v] != slot[prog.inputs[into]]: fixups.append((slot[v], slot[prog.inputs[into]]))
    return slot, nxt, fixups


def compile_prog(prog, G):
    for v in prog.nodes: v.slot_released = False
    for v in prog.inputs.values(): v.slot_released = False
    layers = schedule(prog, G)
    slot, nslots, fixups = allocate(prog, layers)
    if fixups:   # append LIN copy layer(s)
        copies = []
        for src, dst in fixups:
            c = Val(prog, LIN, args=[(1, None)]); c.copy = (src, dst); copies.append(c)
        for i in range(0, len(copies), G): layers.append((LIN, copies[i:i + G]))
    table = []
    for kind, ops in layers:
        row = []
        for v in ops:
            if hasattr(v, "copy"):
                row.append(dict(dst=v.copy[1], a=[v.copy[0], ZERO_SLOT, ZERO_SLOT, ZERO_SLOT], neg=0, half=0, sh=0)); continue
            if kind == MUL:
                A, B = v.args
                terms = (A + [(1, None)] * (2 - len(A))) + (B + [(1, None)] * (2 - len(B)))
            else:
                terms = v.args + [(1, None)] * (4 - len(v.args))
            a = [ZERO_SLOT if t[1] is None else slot[t[1]] for t in terms]
            neg = sum((1 << i) for i, t in enumerate(terms) if t[0] < 0)
            row.append(dict(dst=slot[v], a=a, neg=neg, half=1 if v.half else 0, sh=getattr(v, "sh", 0) if kind == LIN else 0))
        while len(row) < G: row.append(dict(dst=DUMP_SLOT, a=[ZERO_SLOT] * 4, neg=0, half=0, sh=0))
        table.append((kind, row))
    outs = {name: (slot[prog.inputs[into]] if into else slot[v]) for name, v, into in prog.outputs}
    ins = {name: slot[v] for name, v in prog.inputs.items()}
    return dict(name=prog.name, G=G, layers=table, nslots=nslots, ins=ins, outs=outs,
                nmul=sum(1 for k, _ in table if k == MUL), nlin=sum(1 for k, _ in table if k == LIN), lin_ops=sum(1 for v in prog.nodes if v.kind == LIN),
                mul_ops=sum(1 for v in prog.nodes if v.kind == MUL))


def run_compiled(c, inputs):
    """Evaluate the slot program with integers mod p."""
    ws = [0] * max(c["nslots"], 2)
    for name, s in c["ins"].items(): ws[s] = inputs[name] % P
    inv2 = pow(2, -1, P)
    for kind, row in c["layers"]:
        rd = [[ws[s] for s in op["a"]] for op in row]        # all reads before all writes
        for op, r in zip(row, rd):
            sg = [(-1 if (op["neg"] >> i) & 1 else 1) for i in range(4)]
            if kind == MUL: val = ((sg[0] * r[0] + sg[1] * r[1]) * (sg[2] * r[2] + sg[3] * r[3])) % P
            else:
                val = ((sg[0] * r[0] + sg[1] * r[1] + sg[2] * r[2] + sg[3] * r[3]) << op["sh"]) % P
                if op["half"]: val = val * inv2 % P
            if op["dst"] != DUMP_SLOT: ws[op["dst"]] = val
        ws[ZERO_SLOT] = 0
    return {name: ws[s] for name, s in c["outs"].items()}


# ------------------------------------------------------------------------------------------------ programs
def f2in(p, name): return F2(p, p.inp(name + "0"), p.inp(name + "1"))
def f2out(p, name, v, into=None):
    p.out(name + "0", v.c0, into=(into + "0") if into else None); p.out(name + "1", v.c1, into=(into + "1") if into else None)


def prog_line_double():
    """T <- 2T in homogeneous projective coordinates + tangent line scaled for P (bls12_381/pairing.hpp line_double)."""
    p = Prog("line_double")
    X, Y, Z = f2in(p, "X"), f2in(p, "Y"), f2in(p, "Z")
    xP, yP = p.inp("xP"), p.inp("yP")
    a = X.mul(Y).half()
    b, c = Y.sqr().mat(), Z.sqr().mat()
    e = c.mul_xi().scale(12).mat()                     # 4(1+u) * 3c
    f = e.scale(3).mat()
    g = (b + f).half()
    h = ((Y + Z).sqr() - (b + c)).mat()
    i = e - b
    j = X.sqr()
    e2 = e.sqr()
    X3 = a.mul((b - f).mat())
    Y3 = g.sqr() - e2.scale(3)
    Z3 = b.mul(h)
    f2out(p, "X", X3, into="X"); f2out(p, "Y", Y3, into="Y"); f2out(p, "Z", Z3, into="Z")
    f2out(p, "L0", i); f2out(p, "L1", j.scale(3).mat().mul_fp(xP)); f2out(p, "L2", (-h).mul_fp(yP))
    return p


def prog_line_add():
    """T <- T + Q (Q affine) + chord line scaled for P (line_add)."""
    p = Prog("line_add")
    X, Y, Z = f2in(p, "X"), f2in(p, "Y"), f2in(p, "Z")
    xP, yP = p.inp("xP"), p.inp("yP")
    qx, qy = f2in(p, "qx"), f2in(p, "qy")
    theta = (Y - qy.mul(Z)).mat(); lam = (X - qx.mul(Z)).mat()
    c, d = theta.sqr(), lam.sqr().mat()
    e, f, g = lam.mul(d).mat(), Z.mul(c.mat()), X.mul(d).mat()
    h = (e + f - g.scale(2)).mat()
    X3 = lam.mul(h)
    Y3 = theta.mul((g - h).mat()) - e.mul(Y)
    Z3 = Z.mul(e)
    j = theta.mul(qx) - lam.mul(qy)
    f2out(p, "X", X3, into="X"); f2out(p, "Y", Y3, into="Y"); f2out(p, "Z", Z3, into="Z")
    f2out(p, "L0", j); f2out(p, "L1", (-theta).mat().mul_fp(xP)); f2out(p, "L2", lam.mul_fp(yP))
    return p


def fin(p, name, deg): return f2in(p, name) if deg == 2 else F1(p, p.inp(name + "0"))
def fout(p, name, v, into=None):
    if isinstance(v, F2): f2out(p, name, v, into)
    else: p.out(name + "0", v.c0, into=(into + "0") if into else None)


def prog_hom_double(deg):
    """T <- 2T on y^2 = x^3 + b (b = 4 on G1, 4(1+u) on G2), homogeneous projective (x = X/Z, y = Y/Z): the
    doubling of bls12_381/pairing.hpp line_double without the line -- product depth 2."""
    p = Prog("g%d_hdbl" % deg)
    X, Y, Z = fin(p, "X", deg), fin(p, "Y", deg), fin(p, "Z", deg)
    fin(p, "qx", deg); fin(p, "qy", deg)                       # resident affine addend: slots reserved, not used here
    a = X.mul(Y).half()
    b, c = Y.sqr().mat(), Z.sqr().mat()
    e = c.mul_xi().scale(12).mat()
    f = e.scale(3).mat()
    g = (b + f).half()
    h = ((Y + Z).sqr() - (b + c)).mat()
    e2 = e.sqr()
    fout(p, "X", a.mul((b - f).mat()), into="X"); fout(p, "Y", g.sqr() - e2.scale(3), into="Y"); fout(p, "Z", b.mul(h), into="Z")
    return p


def prog_hom_add(deg):
    """T <- T + Q, Q affine; lambda (= 0 iff T = +-Q, the exceptional case) is exported so the kernel can flag it."""
    p = Prog("g%d_hadd" % deg)
    X, Y, Z = fin(p, "X", deg), fin(p, "Y", deg), fin(p, "Z", deg)
    qx, qy = fin(p, "qx", deg), fin(p, "qy", deg)
    theta = (Y - qy.mul(Z)).mat(); lam = (X - qx.mul(Z)).mat()
    c, d = theta.sqr(), lam.sqr().mat()
    e, f, g = lam.mul(d).mat(), Z.mul(c.mat()), X.mul(d).mat()
    h = (e + f - g.scale(2)).mat()
    fout(p, "X", lam.mul(h), into="X"); fout(p, "Y", theta.mul((g - h).mat()) - e.mul(Y), into="Y"); fout(p, "Z", Z.mul(e), into="Z")
    fout(p, "lam", lam)
    return p


def prog_hom_cadd(deg):
    """T <- T + Q with BOTH operands homogeneous projective: the complete addition law of Renes-Costello-Batina (2016, Alg. 7,
    a = 0).  No exceptional case: T = +-Q and the point at infinity (0:1:0) on either side are all handled by the same
    straight-line code -- 12 products, depth 2.  b3 = 3b = 12 on G1, 12(1+u) on G2."""
    p = Prog("g%d_cadd" % deg)
    X1, Y1, Z1 = fin(p, "X", deg), fin(p, "Y", deg), fin(p, "Z", deg)
    X2, Y2, Z2 = fin(p, "qx", deg), fin(p, "qy", deg), fin(p, "qz", deg)
    t0, t1, t2 = X1.mul(X2).mat(), Y1.mul(Y2).mat(), Z1.mul(Z2).mat()
    t3 = ((X1 + Y1).mul(X2 + Y2) - (t0 + t1)).mat()            # X1 Y2 + X2 Y1
    t4 = ((Y1 + Z1).mul(Y2 + Z2) - (t1 + t2)).mat()            # Y1 Z2 + Y2 Z1
    t5 = ((X1 + Z1).mul(X2 + Z2) - (t0 + t2)).mat()            # X1 Z2 + X2 Z1
    b3t2, b3t5 = t2.mul_xi().scale(12).mat(), t5.mul_xi().scale(12).mat()
    m, pl, t03 = (t1 - b3t2).mat(), (t1 + b3t2).mat(), t0.scale(3).mat()
    fout(p, "X", t3.mul(m) - t4.mul(b3t5), into="X"); fout(p, "Y", m.mul(pl) + b3t5.mul(t03), into="Y"); fout(p, "Z", pl.mul(t4) + t03.mul(t3), into="Z")
    return p


class F6:
    def __init__(s, c0, c1, c2): s.c0, s.c1, s.c2 = c0, c1, c2
    def __add__(s, o): return F6(s.c0 + o.c0, s.c1 + o.c1, s.c2 + o.c2)
    def __sub__(s, o): return F6(s.c0 - o.c0, s.c1 - o.c1, s.c2 - o.c2)
    def mul_v(s): return F6(s.c2.mul_xi(), s.c0, s.c1)
    def mul_by_01(s, b0, b1):
        v0, v1 = s.c0.mul(b0), s.c1.mul(b1)
        t0 = ((s.c1 + s.c2).mul(b1) - v1).mul_xi() + v0
        t1 = (s.c0 + s.c1).mul(b0 + b1) - v0 - v1
        t2 = (s.c0 + s.c2).mul(b0) - v0 + v1
        return F6(t0, t1, t2)
    def mul_by_1(s, b1): return F6(s.c2.mul(b1).mul_xi(), s.c0.mul(b1), s.c1.mul(b1))
    def mul(s, o):
        v0, v1, v2 = s.c0.mul(o.c0), s.c1.mul(o.c1), s.c2.mul(o.c2)
        t0 = ((s.c1 + s.c2).mul(o.c1 + o.c2) - v1 - v2).mul_xi() + v0
        t1 = (s.c0 + s.c1).mul(o.c0 + o.c1) - v0 - v1 + v2.mul_xi()
        t2 = (s.c0 + s.c2).mul(o.c0 + o.c2) - v0 - v2 + v1
        return F6(t0, t1, t2)


F12_NAMES = ["c00", "c01", "c02", "c10", "c11", "c12"]


def f12in(p, pre): return [f2in(p, pre + n) for n in F12_NAMES]


def prog_acc_014():
    """f <- f * (l0 + l1 v + l4 v w)   (tower.hpp mul_by_014), f loop-carried in place."""
    p = Prog("acc_014")
    f = f12in(p, "f"); l0, l1, l4 = f2in(p, "l0"), f2in(p, "l1"), f2in(p, "l4")
    c0, c1 = F6(*f[:3]), F6(*f[3:])
    aa = c0.mul_by_01(l0, l1)
    bb = c1.mul_by_1(l4)
    s = (c1 + c0).mul_by_01(l0, l1 + l4) - aa - bb
    r0 = bb.mul_v() + aa
    for n, v in zip(F12_NAMES, [r0.c0, r0.c1, r0.c2, s.c0, s.c1, s.c2]): f2out(p, "f" + n, v, into="f" + n)
    return p


def prog_fp12_mul():
    """f <- f * g (dense), f in place."""
    p = Prog("fp12_mul")
    f, g = f12in(p, "f"), f12in(p, "g")
    a0, a1, b0, b1 = F6(*f[:3]), F6(*f[3:]), F6(*g[:3]), F6(*g[3:])
    v0, v1 = a0.mul(b0), a1.mul(b1)
    x = (a0 + a1).mul(b0 + b1) - v0 - v1
    r0 = v0 + v1.mul_v()
    for n, v in zip(F12_NAMES, [r0.c0, r0.c1, r0.c2, x.c0, x.c1, x.c2]): f2out(p, "f" + n, v, into="f" + n)
    return p


# ------------------------------------------------------------------------------------------------ reference formulas (ints) for validation
def f2m(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2a(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2s(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2k(a, k): return (a[0] * k % P, a[1] * k % P)
def f2xi(a): return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)


def ref_line_double(X, Y, Z, xP, yP):
    i2 = pow(2, -1, P)
    a = f2k(f2m(X, Y), i2); b = f2m(Y, Y); c = f2m(Z, Z); e = f2xi(f2k(c, 12)); f = f2k(e, 3); g = f2k(f2a(b, f), i2)
    h = f2s(f2m(f2a(Y, Z), f2a(Y, Z)), f2a(b, c)); i = f2s(e, b); j = f2m(X, X); e2 = f2m(e, e)
    return dict(X=f2m(a, f2s(b, f)), Y=f2s(f2m(g, g), f2k(e2, 3)), Z=f2m(b, h), L0=i, L1=f2k(f2k(j, 3), xP), L2=f2k(f2k(h, -1), yP))


def ref_line_add(X, Y, Z, qx, qy, xP, yP):
    theta = f2s(Y, f2m(qy, Z)); lam = f2s(X, f2m(qx, Z)); c = f2m(theta, theta); d = f2m(lam, lam)
    e = f2m(lam, d); f = f2m(Z, c); g = f2m(X, d); h = f2s(f2a(e, f), f2k(g, 2))
    return dict(X=f2m(lam, h), Y=f2s(f2m(theta, f2s(g, h)), f2m(e, Y)), Z=f2m(Z, e), L0=f2s(f2m(theta, qx), f2m(lam, qy)),
                L1=f2k(f2k(theta, -1), xP), L2=f2k(lam, yP))


def f12_to_poly(c):   # tower (c00,c01,c02,c10,c11,c12) -> coefficients of w^0..w^5  (v = w^2)
    return [c[0], c[3], c[1], c[4], c[2], c[5]]
def poly_to_f12(g): return [g[0], g[2], g[4], g[1], g[3], g[5]]
def f12m(a, b):
    A, B = f12_to_poly(a), f12_to_poly(b); t = [(0, 0)] * 11
    for i in range(6):
        for j in range(6): t[i + j] = f2a(t[i + j], f2m(A[i], B[j]))
    return poly_to_f12([f2a(t[k], f2xi(t[k + 6])) if k < 5 else t[k] for k in range(6)])


def validate():
    rnd = random.Random(7)
    rf = lambda: rnd.randrange(P); rf2 = lambda: (rf(), rf())
    progs = {}
    for G in (16, 8):
        # line double
        c = compile_prog(prog_line_double(), G); progs[("line_double", G)] = c
        X, Y, Z, xP, yP = rf2(), rf2(), rf2(), rf(), rf()
        out = run_compiled(c, dict(X0=X[0], X1=X[1], Y0=Y[0], Y1=Y[1], Z0=Z[0], Z1=Z[1], xP=xP, yP=yP))
        ref = ref_line_double(X, Y, Z, xP, yP)
        for k, v in ref.items(): assert (out[k + "0"], out[k + "1"]) == v, ("line_double", G, k)
        # line add
        c = compile_prog(prog_line_add(), G); progs[("line_add", G)] = c
        qx, qy = rf2(), rf2()
        out = run_compiled(c, dict(X0=X[0], X1=X[1], Y0=Y[0], Y1=Y[1], Z0=Z[0], Z1=Z[1], qx0=qx[0], qx1=qx[1], qy0=qy[0], qy1=qy[1], xP=xP, yP=yP))
        ref = ref_line_add(X, Y, Z, qx, qy, xP, yP)
        for k, v in ref.items(): assert (out[k + "0"], out[k + "1"]) == v, ("line_add", G, k)
        # sparse accumulate and dense product
        f = [rf2() for _ in range(6)]; g = [rf2() for _ in range(6)]; l0, l1, l4 = rf2(), rf2(), rf2()
        inp = {}
        for n, v in zip(F12_NAMES, f): inp["f" + n + "0"], inp["f" + n + "1"] = v
        c = compile_prog(prog_acc_014(), G); progs[("acc_014", G)] = c
        out = run_compiled(c, dict(inp, l00=l0[0], l01=l0[1], l10=l1[0], l11=l1[1], l40=l4[0], l41=l4[1]))
        sparse = [l0, l1, (0, 0), (0, 0), l4, (0, 0)]
        ref = f12m(f, sparse)
        for n, v in zip(F12_NAMES, ref): assert (out["f" + n + "0"], out["f" + n + "1"]) == v, ("acc_014", G, n)
        c = compile_prog(prog_fp12_mul(), G); progs[("fp12_mul", G)] = c
        for n, v in zip(F12_NAMES, g): inp["g" + n + "0"], inp["g" + n + "1"] = v
        out = run_compiled(c, inp)
        ref = f12m(f, g)
        for n, v in zip(F12_NAMES, ref): assert (out["f" + n + "0"], out["f" + n + "1"]) == v, ("fp12_mul", G, n)
        # group-law programs: compare X/Z, Y/Z with the affine chord-tangent law
        for deg in (1, 2):
            if deg == 1:
                fm = lambda a, b: a * b % P; fi = lambda a: pow(a, -1, P); fs = lambda a, b: (a - b) % P; fk = lambda a, k: a * k % P
                bcoef = 4
            else:
                fm, fs, fk = f2m, f2s, f2k
                fi = lambda a: (lambda d: (a[0] * d % P, (-a[1]) * d % P))(pow(a[0] * a[0] + a[1] * a[1], -1, P))
                bcoef = (4, 4)
            fa = (lambda a, b: (a + b) % P) if deg == 1 else f2a
            def rnd_pt():
                while True:   # random affine point on the curve
                    x = rf() if deg == 1 else rf2()
                    y2 = fa(fm(fm(x, x), x), bcoef)
                    if deg == 1:
                        y = pow(y2, (P + 1) // 4, P)
                        if y * y % P == y2: return x, y
                    else:     # sqrt in Fp2 via the norm method (p = 3 mod 4)
                        a0, a1 = y2
                        n = (a0 * a0 + a1 * a1) % P; sn = pow(n, (P + 1) // 4, P)
                        if sn * sn % P != n: continue
                        for sgn in (1, -1):
                            t = (a0 + sgn * sn) * pow(2, -1, P) % P; x0 = pow(t, (P + 1) // 4, P)
                            if x0 * x0 % P == t and x0:
                                x1 = a1 * pow(2 * x0, -1, P) % P
                                if f2m((x0, x1), (x0, x1)) == y2: return x, (x0, x1)
            def aff_add(p1, p2):
                (x1, y1), (x2, y2) = p1, p2
                lam = fm(fk(fm(x1, x1), 3), fi(fk(y1, 2))) if p1 == p2 else fm(fs(y2, y1), fi(fs(x2, x1)))
                x3 = fs(fs(fm(lam, lam), x1), x2); return x3, fs(fm(lam, fs(x1, x3)), y1)
            T, Q = rnd_pt(), rnd_pt()
            z = rf() if deg == 1 else rf2()
            Th = (fm(T[0], z), fm(T[1], z), z)                # homogeneous representative of T
            def pack(names, vals):
                d = {}
                for nm, v in zip(names, vals):
                    if deg == 1: d[nm + "0"] = v
                    else: d[nm + "0"], d[nm + "1"] = v
                return d
            def unpack(out, nm): return out[nm + "0"] if deg == 1 else (out[nm + "0"], out[nm + "1"])
            c = compile_prog(prog_hom_double(deg), G); progs[("g%d_hdbl" % deg, G)] = c
            out = run_compiled(c, pack(["X", "Y", "Z", "qx", "qy"], Th + Q))
            zi = fi(unpack(out, "Z")); assert (fm(unpack(out, "X"), zi), fm(unpack(out, "Y"), zi)) == aff_add(T, T), ("hdbl", deg, G)
            c = compile_prog(prog_hom_add(deg), G); progs[("g%d_hadd" % deg, G)] = c
            out = run_compiled(c, pack(["X", "Y", "Z", "qx", "qy"], Th + Q))
            zi = fi(unpack(out, "Z")); assert (fm(unpack(out, "X"), zi), fm(unpack(out, "Y"), zi)) == aff_add(T, Q), ("hadd", deg, G)
            # complete addition: generic, doubling, inverse, and infinity on either side
            c = compile_prog(prog_hom_cadd(deg), G); progs[("g%d_cadd" % deg, G)] = c
            z2 = rf() if deg == 1 else rf2()
            zero = 0 if deg == 1 else (0, 0); one = 1 if deg == 1 else (1, 0)
            fneg = (lambda a: (-a) % P) if deg == 1 else (lambda a: ((-a[0]) % P, (-a[1]) % P))
            hom = lambda pt, zz: (fm(pt[0], zz), fm(pt[1], zz), zz)
            inf = (zero, fm(one, z2), zero)
            names = ["X", "Y", "Z", "qx", "qy", "qz"]
            def aff_of(out):
                Zo = unpack(out, "Z")
                if Zo == zero: return None
                zi = fi(Zo); return fm(unpack(out, "X"), zi), fm(unpack(out, "Y"), zi)
            assert aff_of(run_compiled(c, pack(names, Th + hom(Q, z2)))) == aff_add(T, Q), ("cadd", deg, G)
            assert aff_of(run_compiled(c, pack(names, Th + hom(T, z2)))) == aff_add(T, T), ("cadd dbl", deg, G)
            assert aff_of(run_compiled(c, pack(names, Th + hom((T[0], fneg(T[1])), z2)))) is None, ("cadd inverse", deg, G)
            assert aff_of(run_compiled(c, pack(names, inf + hom(Q, z2)))) == Q, ("cadd inf + Q", deg, G)
            assert aff_of(run_compiled(c, pack(names, Th + inf))) == T, ("cadd T + inf", deg, G)
            assert aff_of(run_compiled(c, pack(names, inf + inf))) is None, ("cadd inf + inf", deg, G)
            o2 = run_compiled(c, pack(names, inf + inf)); assert unpack(o2, "Y") != zero     # stays a valid representative of infinity
            # the doubling program maps infinity to infinity
            cd = progs[("g%d_hdbl" % deg, G)]
            o2 = run_compiled(cd, pack(["X", "Y", "Z", "qx", "qy"], inf + Q)); assert unpack(o2, "Z") == zero and unpack(o2, "X") == zero and unpack(o2, "Y") != zero
    return progs


def emit(progs, path):
    w = []
    w.append("// GENERATED by tools/vmgen.py -- do not edit.  Layer tables of the lane-parallel field VM (vm.hpp).")
    w.append("// op = {dst, a0, a1, a2, a3, flags}: flags bits 0-3 = negate term i, bit 4 = halve, bits 5-6 = left shift (LIN only).")
    w.append("#pragma once\nnamespace ripp { namespace vmprog {")
    for (name, G), c in sorted(progs.items()):
        tag = f"{name}_g{G}"
        w.append(f"// {tag}: {c['mul_ops']} Fp products in {c['nmul']} MUL layers + {c['nlin']} LIN layers, {c['nslots']} slots; lane utilisation {c['mul_ops'] / max(1, c['nmul'] * G):.0%}")
        w.append(f"constexpr int {tag}_nlayers = {len(c['layers'])}, {tag}_nslots = {c['nslots']};")
        w.append(f"__device__ const unsigned char {tag}_kind[{len(c['layers'])}] = {{" + ", ".join(str(k) for k, _ in c["layers"]) + "};")
        rows = []
        for _, row in c["layers"]:
            for op in row: rows.append("{%d,%d,%d,%d,%d,%d}" % (op["dst"], op["a"][0], op["a"][1], op["a"][2], op["a"][3], op["neg"] | (op["half"] << 4) | (op["sh"] << 5)))
        w.append(f"__device__ const VmOp {tag}_ops[{len(rows)}] = {{" + ",".join(rows) + "};")
        w.append(f"__device__ const unsigned char {tag}_in[{len(c['ins'])}] = {{" + ", ".join(str(x) for x in c["ins"].values()) + "};   // declaration order")
        w.append(f"__device__ const unsigned char {tag}_out[{len(c['outs'])}] = {{" + ", ".join(str(x) for x in c["outs"].values()) + "};")
        for nm, s in c["ins"].items(): w.append(f"constexpr int {tag}_in_{nm} = {s};")
        for nm, s in c["outs"].items(): w.append(f"constexpr int {tag}_out_{nm} = {s};")
    w.append("} }")
    open(path, "w").write("\n".join(w) + "\n")


if __name__ == "__main__":
    progs = validate()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emit(progs, os.path.join(root, "ripp_amd", "csrc", "vm_programs.inc"))
    for (name, G), c in sorted(progs.items()):
        print(f"{name:12s} G={G:2d}: {c['mul_ops']:3d} products in {c['nmul']:2d} MUL layers, {c['lin_ops']:3d} linear ops in {c['nlin']:2d} LIN layers, {c['nslots']:3d} slots, mul util {c['mul_ops'] / max(1, c['nmul'] * G):.0%}")
