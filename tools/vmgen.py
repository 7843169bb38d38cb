#!/usr/bin/env python3
"""Static scheduler for the lane-parallel field VM (ripp_amd/csrc/vm.hpp), second form.

Idea: G lanes of a wave cooperate on ONE element (a (P,Q) pair, a fold term, an Fp12 accumulator).  All Fp values of the element live
in a per-element LDS workspace ("slots", 14 limbs of 28 bits each: fq28.hpp); a program is a sequence of LAYERS, and in one layer
every lane executes one operation of the same kind:

    MUL   slot[dst] = (+-slot[a0] +-slot[a1]) * (+-slot[a2] +-slot[a3])          (Montgomery product)
    LIN   slot[dst] = sum_t coef[t] * slot[s[t]]   (<= TMAX terms, small signed integer coefficients)

The carry-free radix makes a LIN with ARBITRARY small coefficients one pass of multiply-adds into 64-bit columns, so the chains of
+-1 additions, doublings and halvings the first form needed between two product layers (7 of the 9 layers of a Miller doubling step)
collapse into ONE layer: doubling step 9 -> 4 layers, complete G2 addition 12 -> 7.  Halvings are gone: the projective formulas are
rescaled by powers of two instead (same point; the lines change by factors in Fp, which the final exponentiation kills).

Bounds (units of p) are tracked here and asserted: a MUL result is < 2p; a LIN result is < (sum of |coef| * bound + bias) p and is
either left as it is ("light", <= LIGHT_MAX p) or reduced below 2p ("heavy": one quotient estimate + one more pass); MUL operands
are sums of at most two slots, a negated term is K17 - x with K17 = 17p >= any slot value.

Formulas are written once in a tiny DSL over lazy linear forms; this file list-schedules the resulting DAG into layers for a given
group size G, allocates workspace slots by liveness, EVALUATES the schedule with Python integers against the plain formulas (so a
wrong schedule never reaches the GPU), and emits the tables as a C++ header.

Run:  python tools/vmgen.py   ->  ripp_amd/csrc/vm_programs.inc
"""
import os
import random
import sys

# ---- curve parameters: the formulas below are written once and specialised by these ----------------------------------------------------
#   beta: u^2 in Fp2;  xi: the Fp6 non-residue (w^6 = xi);  twist: M (b' = b xi) or D (b' = b / xi);  b: the G1 curve coefficient;
#   s: projective rescaling that keeps 3 b' times a value a SMALL-INTEGER linear form (BLS12-377: b' = 1/u = -u/5, so everything that meets
#      3 b' c is carried times 5 -- the same projective point, lines scaled by an element of Fp)
_X377 = 0x8508C00000000001
_R377 = _X377**4 - _X377**2 + 1
CURVES = {
    "bls12_381": dict(P=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, beta=-1, xi=(1, 1), twist="M", b=4, s=1,
                      header=os.path.join("ripp_amd", "csrc", "vm_programs.inc"),
                      g1=(0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
                          0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
                      g2=((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
                           0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
                          (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
                           0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))),
    "bls12_377": dict(P=(_X377 - 1) ** 2 * _R377 // 3 + _X377, beta=-5, xi=(0, 1), twist="D", b=1, s=5,
                      header=os.path.join("ripp_amd", "csrc", "bls12_377", "vm_programs.inc"),
                      g1=(81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695,
                          241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030),
                      g2=((233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
                           140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118),
                          (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
                           149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491))),
}
CURVE = CURVES["bls12_381"]
P = CURVE["P"]


def set_curve(name):
    global CURVE, P
    CURVE = CURVES[name]; P = CURVE["P"]


MUL, LIN = 0, 1
ZERO_SLOT = 0          # workspace slot 0 always holds 0; slot 1 is a write-only dump for idle lanes
DUMP_SLOT = 1
FIRST_FREE = 2
TMAX = 16              # terms of one LIN op
COEF_MAX = 127
BOUND_IN = 2           # program inputs are < 2p (callers store canonical values or results of earlier programs)
LIGHT_MAX = 16         # a LIN result up to this many p stays unreduced
HEAVY_MAX = 1000       # what the quotient estimate of a heavy LIN covers
NEG_K = 17             # K17 = 17p: the bias of a negated MUL operand term
VMAX = 2500            # fq28.hpp: operand value bounds must multiply to <= VMAX for the product to come out < 2p


# ------------------------------------------------------------------------------------------------ DSL
class Val:
    """A materialised Fp value (an input or the result of one VM op)."""
    _n = 0

    def __init__(self, prog, kind, args=None, name=None):
        self.prog, self.kind, self.args, self.name = prog, kind, args, name
        self.id = Val._n; Val._n += 1
        self.slot = None
        self.heavy = False
        self.nbias = 0
        if kind == "in": self.bound = BOUND_IN
        elif kind == MUL:
            ba = sum((b.bound if sg > 0 else NEG_K) for sg, b in args[0]); bb = sum((b.bound if sg > 0 else NEG_K) for sg, b in args[1])
            for sg, v in args[0] + args[1]: assert v.bound <= LIGHT_MAX
            assert ba * bb <= VMAX, ("MUL operand bounds", ba, bb)
            self.bound = 2
        else:
            assert 1 <= len(args) <= TMAX and all(abs(c) <= COEF_MAX and c != 0 for c, _ in args), args
            pos = sum(c * v.bound for c, v in args if c > 0); neg = sum(-c * v.bound for c, v in args if c < 0)
            self.nbias = neg + 1 if neg else 0
            raw = pos + self.nbias
            assert raw <= HEAVY_MAX, ("LIN bound", raw)
            self.heavy = raw > LIGHT_MAX
            self.bound = 2 if self.heavy else raw

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
    def nterms(self): return len(self.t)
    def is_val(self): return len(self.t) == 1 and list(self.t.values())[0] == 1
    def simple(self):  # encodable as a MUL operand: at most two terms with coefficient +-1
        return len(self.t) <= 2 and all(abs(c) == 1 for c in self.t.values())
    def terms(self): return [(c, k) for k, c in sorted(self.t.items(), key=lambda kv: kv[0].id)]
    def key(self): return tuple((k.id, c) for k, c in sorted(self.t.items(), key=lambda kv: kv[0].id))


class Prog:
    def __init__(self, name):
        self.name = name; self.inputs = {}; self.outputs = []; self.nodes = []; self.cache = {}

    def inp(self, name):
        v = Val(self, "in", name=name); self.inputs[name] = v; return Lin.of(v)

    def materialise(self, L):
        """Return a Lin of ONE Val equal to L, emitting LIN ops (<= TMAX terms each) as needed; identical forms are shared."""
        L = Lin.of(L)
        if L.is_val(): return L
        k = L.key()
        if k in self.cache: return self.cache[k]
        terms = L.terms()
        assert terms, "materialising zero"
        while len(terms) > TMAX:  # fold the first TMAX terms into one value
            v = Val(self, LIN, args=terms[:TMAX]); self.nodes.append(v); terms = [(1, v)] + terms[TMAX:]
        # coefficients beyond COEF_MAX do not occur in these formulas
        v = Val(self, LIN, args=terms); self.nodes.append(v)
        self.cache[k] = Lin.of(v)
        return self.cache[k]

    def operand(self, L):
        L = Lin.of(L)
        return L if L.simple() else self.materialise(L)

    def mul(self, A, B):
        A, B = self.operand(A), self.operand(B)
        ta = [(c, k) for c, k in A.terms()] or [(1, None)]; tb = [(c, k) for c, k in B.terms()] or [(1, None)]
        if ta == [(1, None)] or tb == [(1, None)]: return Lin()
        v = Val(self, MUL, args=(ta, tb)); self.nodes.append(v); return Lin.of(v)

    def out(self, name, L, into=None):
        """Declare an output; `into` = name of the input whose slot it must end up in (loop-carried state)."""
        L = Lin.of(L)
        if not (L.is_val() and list(L.t.keys())[0].kind != "in"): L = self.materialise_out(L)
        v = list(L.t.keys())[0]
        if v.kind == LIN and v.bound > BOUND_IN: v.heavy = True; v.bound = 2        # kernels read outputs back as canonical values: keep them < 2p
        self.outputs.append((name, v, into))

    def materialise_out(self, L):
        if L.is_val():   # an input passed through: copy it with a 1-term LIN
            v = Val(self, LIN, args=L.terms()); self.nodes.append(v); return Lin.of(v)
        return self.materialise(L)


class F2:
    """Fp2 = Fp[u]/(u^2 - beta) over lazy linear forms (beta = -1 on BLS12-381, -5 on BLS12-377)."""

    def __init__(self, p, c0, c1): self.p, self.c0, self.c1 = p, Lin.of(c0), Lin.of(c1)
    def __add__(self, o): return F2(self.p, self.c0 + o.c0, self.c1 + o.c1)
    def __sub__(self, o): return F2(self.p, self.c0 - o.c0, self.c1 - o.c1)
    def __neg__(self): return F2(self.p, -self.c0, -self.c1)
    def scale(self, c): return F2(self.p, self.c0 * c, self.c1 * c)
    def mul_xi(self):
        if CURVE["xi"] == (1, 1): return F2(self.p, self.c0 - self.c1, self.c0 + self.c1)           # * (1 + u), u^2 = -1
        return F2(self.p, self.c1 * CURVE["beta"], self.c0)                                          # * u
    def mul_3b(self):
        """s * 3 b' * self with b' the twist's curve coefficient: 12 (1 + u) * self (s = 1); on the D-type twist b' = 1/u = u / beta, so
        5 * 3 * (c0 + c1 u) u / beta = 15 (c1 + c0 u / beta) = (15 c1, -3 c0)"""
        if CURVE["twist"] == "M": return self.mul_xi().scale(3 * CURVE["b"])
        assert CURVE["xi"] == (0, 1) and CURVE["s"] == -CURVE["beta"] and CURVE["b"] == 1
        return F2(self.p, self.c1 * (3 * CURVE["s"]), self.c0 * -3)
    def opnd(self):
        """Components usable inside Karatsuba sums: each a single value or a +-1 pair that stays simple when the two are added."""
        a = self
        if not (a.c0.simple() and a.c1.simple() and (a.c0 + a.c1).simple() and (a.c0 - a.c1).simple()): a = a.mat()
        return a
    def mul(self, o):                                                                    # Karatsuba, 3 products
        a, b = self.opnd(), o.opnd()
        t0, t1 = self.p.mul(a.c0, b.c0), self.p.mul(a.c1, b.c1)
        m = self.p.mul(a.c0 + a.c1, b.c0 + b.c1)
        return F2(self.p, t0 + t1 * CURVE["beta"], m - t0 - t1)
    def sqr(self):                                                                       # (a0+a1)(a0-a1), 2 a0 a1
        a = self.opnd()
        if CURVE["beta"] != -1:      # a0^2 + beta a1^2 as two squares: (a0 + a1)(a0 + beta a1) would need a materialised operand, i.e. one more layer
            t0, t1 = self.p.mul(a.c0, a.c0), self.p.mul(a.c1, a.c1)
            return F2(self.p, t0 + t1 * CURVE["beta"], self.p.mul(a.c0, a.c1) * 2)
        m = self.p.mul(a.c0, a.c1)
        return F2(self.p, self.p.mul(a.c0 + a.c1, a.c0 - a.c1), m * 2)
    def mul_fp(self, s): return F2(self.p, self.p.mul(self.c0, s), self.p.mul(self.c1, s))
    def mat(self): return F2(self.p, self.p.materialise(self.c0), self.p.materialise(self.c1))


class F1:
    """Fp itself behind the F2 interface (for the G1 programs)."""

    def __init__(self, p, c0): self.p, self.c0 = p, Lin.of(c0)
    def __add__(self, o): return F1(self.p, self.c0 + o.c0)
    def __sub__(self, o): return F1(self.p, self.c0 - o.c0)
    def __neg__(self): return F1(self.p, -self.c0)
    def scale(self, c): return F1(self.p, self.c0 * c)
    def mul_xi(self): return self                               # G1: no twist factor
    def mul_3b(self): return self.scale(3 * CURVE["b"])        # G1: 3 b (no rescaling needed there: s enters the G1 formulas as 1)
    def mul(self, o): return F1(self.p, self.p.mul(self.c0, o.c0))
    def sqr(self): return F1(self.p, self.p.mul(self.c0, self.c0))
    def mat(self): return F1(self.p, self.p.materialise(self.c0))


# ------------------------------------------------------------------------------------------------ scheduling
def srcs_of(v):
    return [t[1] for t in v.args] if v.kind == LIN else [t[1] for t in v.args[0] + v.args[1]]


def schedule(prog, G):
    """List-schedule prog.nodes into homogeneous layers of <= G ops.  Returns list of (kind, [Val])."""
    nodes = prog.nodes
    deps = {v: {s for s in srcs_of(v) if s.kind != "in"} for v in nodes}
    # critical-path priority (longest path to a sink, MUL weighted 4, LIN 1)
    users = {v: [] for v in nodes}
    for v in nodes:
        for s in deps[v]: users[s].append(v)
    prio = {}
    for v in reversed(nodes):
        prio[v] = (4 if v.kind == MUL else 1) + max([prio[u] for u in users[v]], default=0)
    done, layers, remaining = set(), [], list(nodes)
    while remaining:
        # (1) every LIN that is ready, level by level, each level packed into ceil(n/G) layers (heavy ones first: a layer is heavy if any op is)
        while True:
            rl = [v for v in remaining if v.kind == LIN and deps[v] <= done]
            if not rl: break
            rl.sort(key=lambda v: (not v.heavy, -prio[v]))
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
            for s in srcs_of(v): last_use[s] = li
    outvals = {v for _, v, _ in prog.outputs}
    pinned_into = {v: prog.inputs[into] for _, v, into in prog.outputs if into}
    free = []
    live_inputs = set(prog.inputs.values())
    for li, (_, ops) in enumerate(layers):
        # values whose last use was an earlier layer (inputs are never freed: they are the caller's)
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
    assert nslots <= 255
    table = []
    for kind, ops in layers:
        row = []
        for v in ops:
            if kind == MUL:
                A, B = v.args
                terms = (A + [(1, None)] * (2 - len(A))) + (B + [(1, None)] * (2 - len(B)))
                a = [ZERO_SLOT if t[1] is None else slot[t[1]] for t in terms]
                neg = sum((1 << i) for i, t in enumerate(terms) if t[0] < 0)
                row.append(dict(dst=slot[v], a=a, neg=neg))
            else:
                row.append(dict(dst=slot[v], terms=[(c, slot[x]) for c, x in v.args], nbias=v.nbias, heavy=v.heavy))
        while len(row) < G:
            row.append(dict(dst=DUMP_SLOT, a=[ZERO_SLOT] * 4, neg=0) if kind == MUL else dict(dst=DUMP_SLOT, terms=[], nbias=0, heavy=False))
        table.append((kind, row))
    if fixups:   # outputs that could not be written in place: copy layer(s) at the end
        for i in range(0, len(fixups), G):
            row = [dict(dst=dst, terms=[(1, src)], nbias=0, heavy=False) for src, dst in fixups[i:i + G]]
            while len(row) < G: row.append(dict(dst=DUMP_SLOT, terms=[], nbias=0, heavy=False))
            table.append((LIN, row))
    outs = {name: (slot[prog.inputs[into]] if into else slot[v]) for name, v, into in prog.outputs}
    ins = {name: slot[v] for name, v in prog.inputs.items()}
    return dict(name=prog.name, G=G, layers=table, nslots=nslots, ins=ins, outs=outs,
                nmul=sum(1 for k, _ in table if k == MUL), nlin=sum(1 for k, _ in table if k == LIN), lin_ops=sum(1 for v in prog.nodes if v.kind == LIN),
                mul_ops=sum(1 for v in prog.nodes if v.kind == MUL))


def run_compiled(c, inputs):
    """Evaluate the slot program with integers mod p, tracking the UNREDUCED magnitudes the device sees (in units of p)."""
    ws = [0] * max(c["nslots"], 2)
    for name, s in c["ins"].items(): ws[s] = inputs[name] % P
    for kind, row in c["layers"]:
        if kind == MUL:
            rd = [[ws[s] for s in op["a"]] for op in row]        # all reads before all writes
            for op, r in zip(row, rd):
                sg = [(-1 if (op["neg"] >> i) & 1 else 1) for i in range(4)]
                val = ((sg[0] * r[0] + sg[1] * r[1]) * (sg[2] * r[2] + sg[3] * r[3])) % P
                if op["dst"] != DUMP_SLOT: ws[op["dst"]] = val
        else:
            rd = [[ws[s] for _, s in op["terms"]] for op in row]
            for op, r in zip(row, rd):
                val = sum(cf * x for (cf, _), x in zip(op["terms"], r)) + op["nbias"] * P
                assert val >= 0, "bias does not cover the negative terms"
                if op["dst"] != DUMP_SLOT: ws[op["dst"]] = val % P
        ws[ZERO_SLOT] = 0
    return {name: ws[s] for name, s in c["outs"].items()}


# ------------------------------------------------------------------------------------------------ programs
def f2in(p, name): return F2(p, p.inp(name + "0"), p.inp(name + "1"))
def f2out(p, name, v, into=None):
    p.out(name + "0", v.c0, into=(into + "0") if into else None); p.out(name + "1", v.c1, into=(into + "1") if into else None)


def prog_line_double():
    """T <- 2T in homogeneous projective coordinates + tangent line scaled for P: the step of bls12_381/pairing.hpp line_double with the
    new point scaled by 4 (no halvings: X3 = 2 XY (b - f), Y3 = (b + f)^2 - 12 e^2, Z3 = 4 b h, h = 2 Y Z) -- 2 product layers."""
    p = Prog("line_double")
    X, Y, Z = f2in(p, "X"), f2in(p, "Y"), f2in(p, "Z")
    xP, yP = p.inp("xP"), p.inp("yP")
    s = CURVE["s"]                                     # everything that meets e = 3 b' c is carried times s (s = 1: the formulas as written above)
    b, c = Y.sqr(), Z.sqr()
    e = c.mul_3b()                                     # s * 3 b' c   (BLS12-381: 4(1+u) * 3c)
    f = e.scale(3)
    h = Y.mul(Z).scale(2)                              # (Y + Z)^2 - (b + c)
    bs = b.scale(s)
    i = e - bs
    j = X.sqr()
    X3 = X.mul(Y).scale(2 * s).mul(bs - f)             # s^2 x the textbook point
    Y3 = (bs + f).sqr() - e.sqr().scale(12)
    Z3 = b.mul(h).scale(4) if s == 1 else bs.mul(h.scale(4)).scale(s)      # 4 s^2 b h with every coefficient <= COEF_MAX
    f2out(p, "X", X3, into="X"); f2out(p, "Y", Y3, into="Y"); f2out(p, "Z", Z3, into="Z")
    # s x the textbook line (free, xP, yP) = (e - b, 3 X^2 xP, -h yP); line slots: M-type twist (free, xP, yP) at w^0, w^2, w^3,
    # D-type twist (yP, xP, free) at w^0, w^1, w^3
    if CURVE["twist"] == "M": f2out(p, "L0", i); f2out(p, "L1", j.mul_fp(xP).scale(3 * s)); f2out(p, "L2", (-h.mat()).mul_fp(yP).scale(s))
    else: f2out(p, "L0", (-h.mat()).mul_fp(yP).scale(s)); f2out(p, "L1", j.mul_fp(xP).scale(3 * s)); f2out(p, "L2", i)
    return p


def prog_line_add():
    """T <- T + Q (Q affine) + chord line scaled for P (line_add)."""
    p = Prog("line_add")
    X, Y, Z = f2in(p, "X"), f2in(p, "Y"), f2in(p, "Z")
    xP, yP = p.inp("xP"), p.inp("yP")
    qx, qy = f2in(p, "qx"), f2in(p, "qy")
    theta = (Y - qy.mul(Z)).mat(); lam = (X - qx.mul(Z)).mat()
    c, d = theta.sqr(), lam.sqr()
    e, f, g = lam.mul(d), Z.mul(c), X.mul(d)
    h = e + f - g.scale(2)
    X3 = lam.mul(h)
    Y3 = theta.mul(g - h) - e.mul(Y)
    Z3 = Z.mul(e)
    j = theta.mul(qx) - lam.mul(qy)
    f2out(p, "X", X3, into="X"); f2out(p, "Y", Y3, into="Y"); f2out(p, "Z", Z3, into="Z")
    if CURVE["twist"] == "M": f2out(p, "L0", j); f2out(p, "L1", (-theta).mul_fp(xP)); f2out(p, "L2", lam.mul_fp(yP))
    else: f2out(p, "L0", lam.mul_fp(yP)); f2out(p, "L1", (-theta).mul_fp(xP)); f2out(p, "L2", j)
    return p


def fin(p, name, deg): return f2in(p, name) if deg == 2 else F1(p, p.inp(name + "0"))
def fout(p, name, v, into=None):
    if isinstance(v, F2): f2out(p, name, v, into)
    else: p.out(name + "0", v.c0, into=(into + "0") if into else None)


def prog_hom_double(deg):
    """T <- 2T on y^2 = x^3 + b (b = 4 on G1, 4(1+u) on G2), homogeneous projective (x = X/Z, y = Y/Z): the doubling of line_double
    without the line -- product depth 2."""
    p = Prog("g%d_hdbl" % deg)
    X, Y, Z = fin(p, "X", deg), fin(p, "Y", deg), fin(p, "Z", deg)
    fin(p, "qx", deg); fin(p, "qy", deg)                       # resident affine addend: slots reserved, not used here
    s = CURVE["s"] if deg == 2 else 1
    b, c = Y.sqr(), Z.sqr()
    e = c.mul_3b()
    f = e.scale(3)
    h = Y.mul(Z).scale(2)
    bs = b.scale(s)
    fout(p, "X", X.mul(Y).scale(2 * s).mul(bs - f), into="X"); fout(p, "Y", (bs + f).sqr() - e.sqr().scale(12), into="Y"); fout(p, "Z", b.mul(h).scale(4) if s == 1 else bs.mul(h.scale(4)).scale(s), into="Z")
    return p


def prog_hom_cadd(deg):
    """T <- T + Q with BOTH operands homogeneous projective: the complete addition law of Renes-Costello-Batina (2016, Alg. 7,
    a = 0).  No exceptional case: T = +-Q and the point at infinity (0:1:0) on either side are all handled by the same
    straight-line code.  The cross terms are taken as two products each (no operand sums, hence no linear layer before the first
    products): 15 products, depth 2.  b3 = 3b = 12 on G1, 12(1+u) on G2."""
    p = Prog("g%d_cadd" % deg)
    X1, Y1, Z1 = fin(p, "X", deg), fin(p, "Y", deg), fin(p, "Z", deg)
    X2, Y2, Z2 = fin(p, "qx", deg), fin(p, "qy", deg), fin(p, "qz", deg)
    t0, t1, t2 = X1.mul(X2), Y1.mul(Y2), Z1.mul(Z2)
    t3 = X1.mul(Y2) + X2.mul(Y1)
    t4 = Y1.mul(Z2) + Y2.mul(Z1)
    t5 = X1.mul(Z2) + X2.mul(Z1)
    s = CURVE["s"] if deg == 2 else 1                         # with B = s b3: (X3, Y3, Z3) s^2 = (s t3 (s t1 - B t2) - s t4 B t5, (s t1 - B t2)(s t1 + B t2) + B t5 3 s t0, ...)
    b3t2, b3t5 = t2.mul_3b(), t5.mul_3b()
    t1s, t3s, t4s = t1.scale(s), t3.scale(s), t4.scale(s)
    m, pl, t03 = t1s - b3t2, t1s + b3t2, t0.scale(3 * s)
    fout(p, "X", t3s.mul(m) - t4s.mul(b3t5), into="X"); fout(p, "Y", m.mul(pl) + b3t5.mul(t03), into="Y"); fout(p, "Z", pl.mul(t4s) + t03.mul(t3s), into="Z")
    return p


class F6:
    def __init__(s, c0, c1, c2): s.c0, s.c1, s.c2 = c0, c1, c2
    def __add__(s, o): return F6(s.c0 + o.c0, s.c1 + o.c1, s.c2 + o.c2)
    def __sub__(s, o): return F6(s.c0 - o.c0, s.c1 - o.c1, s.c2 - o.c2)
    def mul_v(s): return F6(s.c2.mul_xi(), s.c0, s.c1)
    def mul(s, o):
        v0, v1, v2 = s.c0.mul(o.c0), s.c1.mul(o.c1), s.c2.mul(o.c2)
        t0 = ((s.c1 + s.c2).mul(o.c1 + o.c2) - v1 - v2).mul_xi() + v0
        t1 = (s.c0 + s.c1).mul(o.c0 + o.c1) - v0 - v1 + v2.mul_xi()
        t2 = (s.c0 + s.c2).mul(o.c0 + o.c2) - v0 - v2 + v1
        return F6(t0, t1, t2)


F12_NAMES = ["c00", "c01", "c02", "c10", "c11", "c12"]


def f12in(p, pre): return [f2in(p, pre + n) for n in F12_NAMES]


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
def f2m(a, b): return ((a[0] * b[0] + CURVE["beta"] * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2a(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2s(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2k(a, k): return (a[0] * k % P, a[1] * k % P)
def f2xi(a): return f2m(a, CURVE["xi"])
def f2inv(a):
    d = pow(a[0] * a[0] - CURVE["beta"] * a[1] * a[1], -1, P); return (a[0] * d % P, (-a[1]) * d % P)
def twist_b(): return f2k(f2xi((1, 0)), CURVE["b"]) if CURVE["twist"] == "M" else f2k(f2inv(CURVE["xi"]), CURVE["b"])


def ref_line_double(X, Y, Z, xP, yP):
    """pairing.hpp line_double, the new point scaled by 4 (same point; the VM avoids the two halvings that way)"""
    i2 = pow(2, -1, P); s = CURVE["s"]
    a = f2k(f2m(X, Y), i2); b = f2m(Y, Y); c = f2m(Z, Z); e = f2m(f2k(c, 3), twist_b()); f = f2k(e, 3); g = f2k(f2a(b, f), i2)
    h = f2s(f2m(f2a(Y, Z), f2a(Y, Z)), f2a(b, c)); i = f2s(e, b); j = f2m(X, X); e2 = f2m(e, e)
    lfree, lx, ly = f2k(i, s), f2k(f2k(j, 3 * s), xP), f2k(f2k(h, -s), yP)                  # the VM carries the point times 4 s^2, the line times s
    L = dict(L0=lfree, L1=lx, L2=ly) if CURVE["twist"] == "M" else dict(L0=ly, L1=lx, L2=lfree)
    return dict(X=f2k(f2m(a, f2s(b, f)), 4 * s * s), Y=f2k(f2s(f2m(g, g), f2k(e2, 3)), 4 * s * s), Z=f2k(f2m(b, h), 4 * s * s), **L)


def ref_line_add(X, Y, Z, qx, qy, xP, yP):
    theta = f2s(Y, f2m(qy, Z)); lam = f2s(X, f2m(qx, Z)); c = f2m(theta, theta); d = f2m(lam, lam)
    e = f2m(lam, d); f = f2m(Z, c); g = f2m(X, d); h = f2s(f2a(e, f), f2k(g, 2))
    lfree, lx, ly = f2s(f2m(theta, qx), f2m(lam, qy)), f2k(f2k(theta, -1), xP), f2k(lam, yP)
    L = dict(L0=lfree, L1=lx, L2=ly) if CURVE["twist"] == "M" else dict(L0=ly, L1=lx, L2=lfree)
    return dict(X=f2m(lam, h), Y=f2s(f2m(theta, f2s(g, h)), f2m(e, Y)), Z=f2m(Z, e), **L)


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
    for G in (16,):
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
        # dense product
        f = [rf2() for _ in range(6)]; g = [rf2() for _ in range(6)]
        inp = {}
        for n, v in zip(F12_NAMES, f): inp["f" + n + "0"], inp["f" + n + "1"] = v
        c = compile_prog(prog_fp12_mul(), G); progs[("fp12_mul", G)] = c
        for n, v in zip(F12_NAMES, g): inp["g" + n + "0"], inp["g" + n + "1"] = v
        out = run_compiled(c, inp)
        ref = f12m(f, g)
        for n, v in zip(F12_NAMES, ref): assert (out["f" + n + "0"], out["f" + n + "1"]) == v, ("fp12_mul", G, n)
        # group-law programs: compare X/Z, Y/Z with the affine chord-tangent law
        for deg in (1, 2):
            if deg == 1:
                fm = lambda a, b: a * b % P; fi = lambda a: pow(a, -1, P); fs = lambda a, b: (a - b) % P; fk = lambda a, k: a * k % P
            else:
                fm, fs, fk, fi = f2m, f2s, f2k, f2inv
            fa = (lambda a, b: (a + b) % P) if deg == 1 else f2a
            def aff_add(p1, p2):
                (x1, y1), (x2, y2) = p1, p2
                lam = fm(fk(fm(x1, x1), 3), fi(fk(y1, 2))) if p1 == p2 else fm(fs(y2, y1), fi(fs(x2, x1)))
                x3 = fs(fs(fm(lam, lam), x1), x2); return x3, fs(fm(lam, fs(x1, x3)), y1)
            def rnd_pt():         # a random multiple of the generator (no square roots: p = 1 mod 4 on BLS12-377)
                gen = CURVE["g1"] if deg == 1 else CURVE["g2"]
                acc = None
                for bit in bin(rnd.randrange(2, 1 << 40))[2:]:
                    if acc is not None: acc = aff_add(acc, acc)
                    if bit == "1": acc = gen if acc is None else aff_add(acc, gen)
                return acc
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
    w.append("// GENERATED by tools/vmgen.py -- do not edit.  Layer tables of the lane-parallel field VM (vm.hpp), %s." % [k for k, v in CURVES.items() if v is CURVE][0].upper().replace("_", "-"))
    w.append("// kind[l]: 0 = MUL layer; otherwise LIN layer: bits 0-4 = terms walked by the layer (max over its ops), bit 6 = heavy (results reduced below 2p).")
    w.append("// op = {dst, flags, nbias, s[16], c[16]}: MUL uses s[0..3] (second term of an operand absent <=> slot 0) and flags bits 0-3 = negate term i;")
    w.append("// LIN: dst = sum_t c[t] * slot[s[t]] + nbias * p  (unused terms: slot 0, coefficient 0).")
    w.append("#pragma once\nnamespace ripp { namespace vmprog {")
    for (name, G), c in sorted(progs.items()):
        tag = f"{name}_g{G}"
        w.append(f"// {tag}: {c['mul_ops']} Fp products in {c['nmul']} MUL layers + {c['lin_ops']} linear ops in {c['nlin']} LIN layers, {c['nslots']} slots; lane utilisation {c['mul_ops'] / max(1, c['nmul'] * G):.0%}")
        w.append(f"constexpr int {tag}_nlayers = {len(c['layers'])}, {tag}_nslots = {c['nslots']};")
        kinds = []
        rows = []
        for kind, row in c["layers"]:
            if kind == MUL:
                kinds.append(0)
                for op in row: rows.append("{%d,%d,0,{%d,%d,%d,%d},{0}}" % (op["dst"], op["neg"], op["a"][0], op["a"][1], op["a"][2], op["a"][3]))
            else:
                tmax = max([len(op["terms"]) for op in row] + [1]); heavy = any(op["heavy"] for op in row)
                kinds.append(0x80 | (0x40 if heavy else 0) | tmax)
                for op in row:
                    sl = [t[1] for t in op["terms"]]; cf = [t[0] for t in op["terms"]]
                    rows.append("{%d,0,%d,{%s},{%s}}" % (op["dst"], op["nbias"], ",".join(map(str, sl)) or "0", ",".join(map(str, cf)) or "0"))
        w.append(f"__device__ const unsigned char {tag}_kind[{len(kinds)}] = {{" + ", ".join(str(k) for k in kinds) + "};")
        w.append(f"__device__ const VmOp {tag}_ops[{len(rows)}] = {{" + ",".join(rows) + "};")
        w.append(f"__device__ const unsigned char {tag}_in[{len(c['ins'])}] = {{" + ", ".join(str(x) for x in c["ins"].values()) + "};   // declaration order")
        w.append(f"__device__ const unsigned char {tag}_out[{len(c['outs'])}] = {{" + ", ".join(str(x) for x in c["outs"].values()) + "};")
        for nm, s in c["ins"].items(): w.append(f"constexpr int {tag}_in_{nm} = {s};")
        for nm, s in c["outs"].items(): w.append(f"constexpr int {tag}_out_{nm} = {s};")
    w.append("} }")
    open(path, "w").write("\n".join(w) + "\n")


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for curve in CURVES:
        set_curve(curve)
        progs = validate()
        emit(progs, os.path.join(root, CURVE["header"]))
        print(curve)
        for (name, G), c in sorted(progs.items()):
            print(f"  {name:12s} G={G:2d}: {c['mul_ops']:3d} products in {c['nmul']:2d} MUL layers, {c['lin_ops']:3d} linear ops in {c['nlin']:2d} LIN layers, {c['nslots']:3d} slots, mul util {c['mul_ops'] / max(1, c['nmul'] * G):.0%}")
    set_curve("bls12_381")
