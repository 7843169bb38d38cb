"""rust/ripp-hip/src/ffi.rs against include/ripp_hip.h, mechanically (no Rust toolchain in this image: a drifted signature would otherwise
show up on somebody else's machine, as a crash).  An INDEPENDENT parser of both files -- not the generator's (tools/gen_rust_ffi.py):
  * every prototype of the header has a `pub fn` of the same name and vice versa (or sits on the explicit allow-list of unbound hooks);
  * per argument: pointer depth, constness of the pointee, integer width / struct identity; same for the return type;
  * RippStats, RippConfig, RippAggregateProof, RippVerifierSrs, RippGroth16Vk: field by field (name, type, order) against the typedefs;
  * the ABI version constants agree."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "ripp_hip.h")).read()
FFI = open(os.path.join(ROOT, "rust", "ripp-hip", "src", "ffi.rs")).read()

# exports a Rust host has no business calling (none today: the generated block binds everything, test hooks included)
UNBOUND_ALLOWED = set()

C_SCALAR = {"int32_t": "i32", "uint32_t": "u32", "size_t": "usize", "uint64_t": "u64", "double": "f64", "uint8_t": "u8", "char": "c_char", "void": "c_void"}


def c_nocomment(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def rs_nocomment(text):
    return "\n".join(re.sub(r"//.*$", "", ln) for ln in text.splitlines())


def canon_c(ctype, array=False):
    """C type -> (pointer depth, const pointee, base name in Rust spelling)"""
    t = ctype.strip()
    const = bool(re.match(r"const\b", t))
    t = re.sub(r"^const\s+", "", t)
    depth = t.count("*") + (1 if array else 0)
    base = t.replace("*", "").strip()
    if base == "ripp_allgather_fn":
        return (0, False, "RippAllgatherFn")
    if base in C_SCALAR:
        base = C_SCALAR[base]
    else:
        assert base.startswith("ripp_"), ctype
        parts = base[5:].split("_")
        special = {"g1a": "G1A", "g1j": "G1J", "g2a": "G2A", "g2j": "G2J"}
        base = "Ripp" + "".join(special.get(p, p.capitalize()) for p in parts)
    return (depth, const and depth > 0, base)


def canon_rs(rtype):
    t = rtype.strip()
    depth, const = 0, False
    while True:
        m = re.match(r"\*(const|mut)\s+(.*)$", t)
        if not m:
            break
        depth += 1; const = m.group(1) == "const"; t = m.group(2).strip()      # the innermost qualifier wins (the pointee's)
    t = t.replace("core::ffi::", "")
    return (depth, const and depth > 0, t)


def c_prototypes():
    out = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_ \*]*?)\s*\b(ripp_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", c_nocomment(HDR), flags=re.M | re.S):
        args = re.sub(r"\s+", " ", args.strip())
        params = []
        if args != "void":
            for a in args.split(","):
                m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[[0-9]*\])?$", a.strip())
                params.append(canon_c(m.group(1), array=bool(m.group(3))))
        out[name] = (None if ret.strip() == "void" else canon_c(ret), params)
    return out


def rs_prototypes():
    out = {}
    for name, args, ret in re.findall(r"pub fn (ripp_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", rs_nocomment(FFI), flags=re.S):
        params = []
        depth, cur = 0, ""
        for ch in args + ",":                       # split on top-level commas (fn-pointer types contain commas)
            if ch == "," and depth == 0:
                if cur.strip():
                    params.append(canon_rs(cur.split(":", 1)[1]))
                cur = ""
            else:
                depth += ch in "(<"; depth -= ch in ")>"; cur += ch
        out[name] = (canon_rs(ret) if ret else None, params)
    return out


def test_every_export_is_bound_with_the_same_signature():
    c, r = c_prototypes(), rs_prototypes()
    assert len(c) >= 114, len(c)
    missing = sorted(set(c) - set(r) - UNBOUND_ALLOWED)
    extra = sorted(set(r) - set(c))
    assert not missing, f"header exports without a binding in ffi.rs: {missing}"
    assert not extra, f"ffi.rs declares functions the header does not have: {extra}"
    for name in sorted(set(c) & set(r)):
        (cret, cpar), (rret, rpar) = c[name], r[name]
        assert cret == rret, f"{name}: return type {rret} in ffi.rs, {cret} in the header"
        assert len(cpar) == len(rpar), f"{name}: {len(rpar)} parameters in ffi.rs, {len(cpar)} in the header"
        for i, (a, b) in enumerate(zip(cpar, rpar)):
            assert a == b, f"{name}: parameter {i} is {b} in ffi.rs, {a} in the header"


def c_struct_fields(name):
    """fields of `typedef struct { ... } name;` as [(field, canon type)]"""
    src = c_nocomment(HDR)
    m = re.search(r"typedef struct\s*\{([^{}]*)\}\s*%s\s*;" % re.escape(name), src, flags=re.S)
    assert m, name
    fields = []
    for decl in m.group(1).split(";"):
        decl = re.sub(r"\s+", " ", decl.strip())
        if not decl:
            continue
        m2 = re.match(r"((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*)\s*(.*)$", decl)
        base, rest = m2.group(1), m2.group(2)
        for item in rest.split(","):
            item = item.strip()
            stars = item.count("*")
            fields.append((item.replace("*", "").strip(), canon_c(base + "*" * stars)))
    return fields


def rs_struct_fields(name):
    m = re.search(r"pub struct %s\s*\{([^{}]*)\}" % re.escape(name), rs_nocomment(FFI), flags=re.S)      # (these structs hold no nested braces)
    assert m, name
    return [(f, canon_rs(t)) for f, t in re.findall(r"pub\s+([A-Za-z_][A-Za-z0-9_]*)\s*:\s*([^,}]+)", m.group(1))]


def test_structs_match_field_by_field():
    for cname, rname in (("ripp_stats", "RippStats"), ("ripp_config", "RippConfig"), ("ripp_aggregate_proof", "RippAggregateProof"),
                         ("ripp_verifier_srs", "RippVerifierSrs"), ("ripp_groth16_vk", "RippGroth16Vk")):
        c, r = c_struct_fields(cname), rs_struct_fields(rname)
        assert [f for f, _ in c] == [f for f, _ in r], f"{rname}: fields {[f for f, _ in r]} vs header {[f for f, _ in c]}"
        for (f, ct), (_, rt) in zip(c, r):
            assert ct == rt, f"{rname}.{f}: {rt} in ffi.rs, {ct} in the header"


def test_field_element_layouts():
    """limb counts of the POD field / point structs"""
    src = c_nocomment(HDR)
    for cname, rname in (("ripp_fp", "RippFp"), ("ripp_fr", "RippFr")):
        n_c = int(re.search(r"typedef struct\s*\{\s*uint64_t l\[(\d+)\];\s*\}\s*%s;" % cname, src).group(1))
        n_r = int(re.search(r"pub struct %s \{ pub l: \[u64; (\d+)\] \}" % rname, FFI).group(1))
        assert n_c == n_r, (cname, n_c, n_r)
    assert re.search(r"pub struct RippGt \{ pub c: \[RippFp2; 6\] \}", FFI) and re.search(r"typedef struct\s*\{\s*ripp_fp2 c\[6\];\s*\}\s*ripp_gt;", src)


def test_abi_version_constants_agree():
    v_h = int(re.search(r"#define RIPP_ABI_VERSION (\d+)", HDR).group(1))
    v_r = int(re.search(r"pub const RIPP_ABI_VERSION: i32 = (\d+);", FFI).group(1))
    from ripp_amd._lib import RIPP_ABI_VERSION
    assert v_h == v_r == RIPP_ABI_VERSION


def test_generated_block_is_current():
    import subprocess
    import sys
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]).returncode == 0, "run tools/gen_rust_ffi.py"
