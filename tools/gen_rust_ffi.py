#!/usr/bin/env python3
"""Regenerates the `extern "C"` block of rust/ripp-hip/src/ffi.rs from include/ripp_hip.h (everything between the GENERATED markers).

The image has no Rust toolchain, so a signature that drifts from the header would only show on somebody else's machine, as a crash.  The
declarations are therefore not typed by hand: this script maps every prototype of the header to its Rust form, and
tests/test_rust_ffi_signatures_cpu.py -- an independent parser of BOTH files -- checks arity, pointer constness and integer widths of every
`pub fn ripp_*`, and `RippStats` / `RippConfig` / the aggregate structs field by field.

  python tools/gen_rust_ffi.py          # rewrite ffi.rs in place
  python tools/gen_rust_ffi.py --check  # exit 1 if ffi.rs is out of date
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "ripp_hip.h")
FFI = os.path.join(ROOT, "rust", "ripp-hip", "src", "ffi.rs")
BEGIN, END = "    // ---- GENERATED from include/ripp_hip.h by tools/gen_rust_ffi.py: do not edit by hand ----\n", "    // ---- end of the generated block ----\n"

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "size_t": "usize", "uint64_t": "u64", "int64_t": "i64", "double": "f64", "uint8_t": "u8", "char": "core::ffi::c_char", "void": "c_void"}
STRUCTS = {"ripp_fp": "RippFp", "ripp_fr": "RippFr", "ripp_fp2": "RippFp2", "ripp_gt": "RippGt", "ripp_g1a": "RippG1A", "ripp_g1j": "RippG1J", "ripp_g2a": "RippG2A",
           "ripp_g2j": "RippG2J", "ripp_stats": "RippStats", "ripp_config": "RippConfig", "ripp_vec": "RippVec", "ripp_sipp_job": "RippSippJob", "ripp_srs": "RippSrs",
           "ripp_aggregate_proof": "RippAggregateProof", "ripp_verifier_srs": "RippVerifierSrs", "ripp_groth16_vk": "RippGroth16Vk"}


def prototypes(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_ \*]*?)\s*\b(ripp_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        args = re.sub(r"\s+", " ", args.strip())
        out.append((ret.strip(), name, [] if args == "void" else [a.strip() for a in args.split(",")]))
    return out


def rust_type(ctype):
    """'const ripp_g1j*' -> '*const RippG1J'; arrays in parameter position are pointers."""
    ctype = ctype.strip()
    if ctype == "ripp_allgather_fn":
        return "RippAllgatherFn"
    const = ctype.startswith("const ")
    base = ctype[6:] if const else ctype
    stars = base.count("*")
    base = base.replace("*", "").strip()
    r = SCALARS.get(base) or STRUCTS.get(base)
    if r is None:
        raise SystemExit(f"gen_rust_ffi: no Rust counterpart for C type '{ctype}'")
    if stars == 0:
        assert base != "void"
        return r
    t = ("*const " if const else "*mut ") + r
    for _ in range(stars - 1):
        t = "*mut " + t
    return t


def param(decl):
    m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[[0-9]*\])?$", decl)
    ctype, name, arr = m.group(1).strip(), m.group(2), m.group(3)
    if arr:
        ctype += "*"
    if name in ("in", "type", "ref", "match", "move", "box", "fn", "loop", "use"):
        name += "_"
    return name, rust_type(ctype)


def generate():
    lines = []
    for ret, name, args in prototypes(open(HDR).read()):
        ps = ", ".join(f"{n}: {t}" for n, t in (param(a) for a in args))
        r = "" if ret == "void" else " -> " + rust_type(ret)
        lines.append(f"    pub fn {name}({ps}){r};\n")
    return "".join(lines)


def main():
    src = open(FFI).read()
    i, j = src.index(BEGIN), src.index(END)
    new = src[: i + len(BEGIN)] + generate() + src[j:]
    if "--check" in sys.argv:
        sys.exit(0 if new == src else 1)
    open(FFI, "w").write(new)


if __name__ == "__main__":
    main()
