#!/usr/bin/env python3
"""Generates blake2s_x64.S: BLAKE2s compression of N full (non-final) blocks, scalar x86-64, hand-allocated.

Why: the statement hash of SIPP::prove (sipp/src/lib.rs:56-59; 336 MB at n = 2^20) is sequential and is 73 % of the single-GPU step.  The G
function's dependency chain (b -> a -> d -> c -> b, six 1-cycle operations per half) allows 240 cycles per block; compilers reach ~281 on Zen 5
because the sixteen state words, the stack pointer and a message pointer do not fit sixteen registers and their spill code lands on the chain.
This form keeps fifteen words' worth of state in registers by construction:
  * b, c, d rows: twelve registers; a0, a1: two registers; a2 and a3 live in STACK SLOTS and pass through ONE shared temporary while their G runs
    (a word of the `a` row is idle from the second `xor d, a` of one half-round to the first `add a, b` of the next: 5 cycles of slack for the
    store -> load round trip, which rsp-relative memory renaming makes ~free on Zen);
  * the message block is copied to the stack once per block (four 16-byte moves), so message words are rsp-relative memory operands and no
    register holds a pointer during the rounds; h, the input pointer, the counter and t live in the frame as well;
  * `(a + m) + b`: the message word is added before b arrives, one dependent add on the chain.
Variants (argv[1]): "seq" = the two stack-resident G's after the two register G's in program order; "lock" = three G's in lockstep, the fourth after.
"""
import sys

SIGMA = [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
         [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
         [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
         [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
         [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]
IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
A = ["eax", "ebx", None, None]                      # a2, a3: stack slots through T
T = "ecx"
B = ["edx", "esi", "edi", "ebp"]
C = ["r8d", "r9d", "r10d", "r11d"]
D = ["r12d", "r13d", "r14d", "r15d"]
# frame (rsp-relative): 0..63 message copy, 64 a2, 68 a3, 72..103 h[8], 104 in, 112 n, 120 t, 128 h pointer
M, A2, A3, H, IN, N, TT, HP, FRAME = 0, 64, 68, 72, 104, 112, 120, 128, 136


XMM_HOME = False          # argv option "xmm": a2 / a3 live in xmm4 / xmm5 (movd both ways) instead of stack slots


def ld_a(k): return f"movd {T}, xmm{k + 2}" if XMM_HOME else f"mov {T}, [rsp+{A2 if k == 2 else A3}]"
def st_a(k): return f"movd xmm{k + 2}, {T}" if XMM_HOME else f"mov [rsp+{A2 if k == 2 else A3}], {T}"


def g_ops(k, half, x, y, immediate=False):
    """the 14 (16 for stack-resident a) instructions of one G as a list; k = G index 0..3.
    immediate: a stack-resident a goes back to its slot right after EACH update (T is free again at once, so such G's can be interleaved)"""
    a = A[k]
    if half == 0:
        b, c, d = B[k], C[k], D[k]
    else:
        b, c, d = B[(k + 1) % 4], C[(k + 2) % 4], D[(k + 3) % 4]
    ops = []
    mem = a is None
    slot = A2 if k == 2 else A3
    if mem:
        a = T
        ops.append(ld_a(k))
    ops += [f"add {a}, [rsp+{M + 4 * x}]", f"add {a}, {b}"]
    if mem and immediate:
        ops.append(st_a(k))
    ops += [f"xor {d}, {a}", f"ror {d}, 16", f"add {c}, {d}", f"xor {b}, {c}", f"ror {b}, 12"]
    if mem and immediate:
        ops.append(ld_a(k))
    ops += [f"add {a}, [rsp+{M + 4 * y}]", f"add {a}, {b}"]
    if mem and immediate:
        ops.append(st_a(k))
    ops += [f"xor {d}, {a}", f"ror {d}, 8"]
    if mem and not immediate:
        ops.append(st_a(k))
    ops += [f"add {c}, {d}", f"xor {b}, {c}", f"ror {b}, 7"]
    return ops


def interleave(lists):
    out = []
    while any(lists):
        for l in lists:
            if l:
                out.append(l.pop(0))
    return out


def interleave_units(lists, unit):
    """lockstep in units of `unit` instructions per G (keeps an immediate G's load-add-add-store / its T lifetime contiguous)"""
    out = []
    while any(lists):
        for l in lists:
            for _ in range(unit):
                if l:
                    out.append(l.pop(0))
    return out


def half_round(r, half, variant):
    s = SIGMA[r][8 * half: 8 * half + 8]
    if variant == "lock4m":                         # all four in lockstep; a2 / a3 return to their slots after every update (T lives 4 instructions)
        g = [g_ops(k, half, s[2 * k], s[2 * k + 1], immediate=True) for k in range(4)]
        # phases of a G: [a-update][d][c][b][a-update][d][c][b]; emit phase by phase across the four G's, the stack-resident ones first
        def phases(ops, mem):
            cut = [5, 1, 1, 2, 5, 1, 1, 2] if mem else [3, 1, 1, 2, 3, 1, 1, 2]      # (the xor that consumes T stays with the update: T is dead at every phase boundary)
            out, i = [], 0
            for c in cut:
                out.append(ops[i:i + c]); i += c
            assert i == len(ops), (i, len(ops))
            return out
        ph = [phases(g[k], A[k] is None) for k in range(4)]
        out = []
        for p in range(8):
            for k in (3, 2, 0, 1):
                out += ph[k][p]
        return out
    g = [g_ops(k, half, s[2 * k], s[2 * k + 1]) for k in range(4)]
    if variant == "seq":
        return interleave([g[0], g[1]]) + g[2] + g[3]
    if variant == "lock":
        return interleave([g[0], g[1], g[2]]) + g[3]
    if variant == "lock3first":                     # the lone stack-resident G FIRST in program order (oldest-first pick favours it), then three in lockstep
        return g[3] + interleave([g[0], g[1], g[2]])
    if variant == "l3f_u2":                          # as lock3first, the lockstep in units of two instructions per G
        return g[3] + interleave_units([g[0], g[1], g[2]], 2)
    if variant == "l3f_u3":
        return g[3] + interleave_units([g[0], g[1], g[2]], 3)
    if variant == "l3f_201":                         # as lock3first, the stack-resident G2 leading the lockstep
        return g[3] + interleave([g[2], g[0], g[1]])
    if variant == "l3f_half":                        # G3's first half-G, the other three's first half-G's in lockstep, then the second halves likewise
        h3 = len(g[3]) // 2
        a3, b3 = g[3][:8], g[3][8:]
        firsts = [x[:7] for x in (g[0], g[1])] + [g[2][:8]]
        seconds = [x[7:] for x in (g[0], g[1])] + [g[2][8:]]
        return a3 + interleave(firsts) + b3 + interleave(seconds)
    if variant == "g32first":                       # both stack-resident G's first, then the two register G's in lockstep
        return g[3] + g[2] + interleave([g[0], g[1]])
    raise SystemExit("variant?")


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "lock"
    opts = set(sys.argv[2:])
    keep_h, early = "keep_h" in opts, "early_copy" in opts
    global XMM_HOME
    XMM_HOME = "xmm" in opts
    name = next((o[5:] for o in opts if o.startswith("name=")), None)
    opts = {o for o in opts if not o.startswith("name=")}
    name = name or "blake2s_blocks_" + "_".join([variant] + sorted(opts))
    copy_msg = lambda reg: [f"movdqu xmm0, [{reg}]", f"movdqu xmm1, [{reg}+16]", f"movdqu xmm2, [{reg}+32]", f"movdqu xmm3, [{reg}+48]",
                            f"movdqu [rsp+{M}], xmm0", f"movdqu [rsp+{M + 16}], xmm1", f"movdqu [rsp+{M + 32}], xmm2", f"movdqu [rsp+{M + 48}], xmm3"]
    load_ab = [f"mov {A[0]}, [rsp+{H}]", f"mov {A[1]}, [rsp+{H + 4}]", f"mov {T}, [rsp+{H + 8}]", st_a(2), f"mov {T}, [rsp+{H + 12}]", st_a(3)] + \
              [f"mov {B[i]}, [rsp+{H + 16 + 4 * i}]" for i in range(4)]
    o = [f"# GENERATED by tools/ubench/gen_blake2s_x64.py {' '.join(sys.argv[1:])} -- do not edit; see that script for the design and tools/ubench/blake2s_bench.cpp for the measurements",
         ".intel_syntax noprefix", ".text", f".globl {name}", f".hidden {name}", f".type {name}, @function", f"{name}:",
         "# void f(uint32_t h[8] /*rdi*/, const uint8_t* in /*rsi*/, size_t nblocks /*rdx*/, uint64_t t /*rcx: bytes hashed before the first block*/)",
         "# early_copy forms read 64 bytes BEYOND the last block they process (the next block's message is staged during the current one)",
         "push rbx", "push rbp", "push r12", "push r13", "push r14", "push r15", f"sub rsp, {FRAME}",
         "test rdx, rdx", "jz 9f",
         f"mov [rsp+{HP}], rdi", f"mov [rsp+{IN}], rsi", f"mov [rsp+{N}], rdx", f"mov [rsp+{TT}], rcx"]
    for i in range(0, 8, 2):
        o += [f"mov rax, [rdi+{4 * i}]", f"mov [rsp+{H + 4 * i}], rax"]
    if early:
        o += copy_msg("rsi") + ["add rsi, 64", f"mov [rsp+{IN}], rsi"]
    if keep_h:
        o += load_ab
    o += [".p2align 6", "1:"]
    if not early:
        o += [f"mov rcx, [rsp+{IN}]"] + copy_msg("rcx") + ["add rcx, 64", f"mov [rsp+{IN}], rcx"]
    o += [f"mov rcx, [rsp+{TT}]", "add rcx, 64", f"mov [rsp+{TT}], rcx"]
    o += [f"mov {D[0]}, ecx", f"xor {D[0]}, {IV[4]}", "shr rcx, 32", f"mov {D[1]}, ecx", f"xor {D[1]}, {IV[5]}", f"mov {D[2]}, {IV[6]}", f"mov {D[3]}, {IV[7]}"]
    if not keep_h:
        o += load_ab
    o += [f"mov {C[i]}, {IV[i]}" for i in range(4)]
    for r in range(10):
        for half in (0, 1):
            o += half_round(r, half, variant)
    if early:       # stage the NEXT block's message behind the last round's loads (program order keeps them on the old words); rcx is free here
        o += [f"mov rcx, [rsp+{IN}]"] + copy_msg("rcx") + ["add rcx, 64", f"mov [rsp+{IN}], rcx"]
    if keep_h:      # new h in the registers the next block starts from; the frame copy is updated off the chain
        o += [x for k in (0, 1) for x in (f"xor {A[k]}, {C[k]}", f"xor {A[k]}, [rsp+{H + 4 * k}]", f"mov [rsp+{H + 4 * k}], {A[k]}")]
        for k, slot in ((2, A2), (3, A3)):
            o += [ld_a(k), f"xor {T}, {C[k]}", f"xor {T}, [rsp+{H + 4 * k}]", st_a(k), f"mov [rsp+{H + 4 * k}], {T}"]
        o += [x for i in range(4) for x in (f"xor {B[i]}, {D[i]}", f"xor {B[i]}, [rsp+{H + 16 + 4 * i}]", f"mov [rsp+{H + 16 + 4 * i}], {B[i]}")]
    else:
        o += [f"xor {A[0]}, {C[0]}", f"xor [rsp+{H}], {A[0]}", f"xor {A[1]}, {C[1]}", f"xor [rsp+{H + 4}], {A[1]}",
              ld_a(2), f"xor {T}, {C[2]}", f"xor [rsp+{H + 8}], {T}", ld_a(3), f"xor {T}, {C[3]}", f"xor [rsp+{H + 12}], {T}"]
        o += [x for i in range(4) for x in (f"xor {B[i]}, {D[i]}", f"xor [rsp+{H + 16 + 4 * i}], {B[i]}")]
    o += [f"dec qword ptr [rsp+{N}]", "jnz 1b"]
    o += [f"mov rdi, [rsp+{HP}]"]
    for i in range(0, 8, 2):
        o += [f"mov rax, [rsp+{H + 4 * i}]", f"mov [rdi+{4 * i}], rax"]
    o += ["9:", f"add rsp, {FRAME}", "pop r15", "pop r14", "pop r13", "pop r12", "pop rbp", "pop rbx", "ret", f".size {name}, .-{name}", '.section .note.GNU-stack,"",@progbits']
    import re
    if "rorx" in opts:          # BMI2 rotate: no flags written
        o = [re.sub(r"^ror (\w+), (\d+)$", lambda m: f"rorx {m.group(1)}, {m.group(1)}, {m.group(2)}", x) for x in o]
    if "lea" in opts:           # register-register adds as lea: no flags written
        def lea(x):
            m = re.match(r"^add (e\w+|r\d+d), (e\w+|r\d+d)$", x)
            if not m:
                return x
            r64 = lambda r: ("r" + r[1:]) if r.startswith("e") else r[:-1]
            return f"lea {m.group(1)}, [{r64(m.group(1))}+{r64(m.group(2))}]"
        o = [lea(x) for x in o]
    print("\n".join(("    " + x) if not x.endswith(":") and not x.startswith(".") and not x.startswith("#") else x for x in o))


if __name__ == "__main__":
    main()
