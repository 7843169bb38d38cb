#!/usr/bin/env python3
"""Compare the golden vectors of this repository with a dump produced by REAL arkworks (rust/ripp-hip/examples/dump_kat.rs).

  python3 tools/compare_kat.py tests/golden/bls12_381_vectors.json kat_arkworks.json

Every key the dump holds must equal the golden file's value (hex strings compared case-insensitively, integers as integers).
Exit status 0 = the oracle's conventions are pinned by arkworks itself; the list printed otherwise names each [ark-mem] convention
that differs."""
import json
import sys


def norm(v):
    if isinstance(v, str):
        s = v.lower()
        if s.startswith("0x"):
            return int(s, 16)
        return s
    if isinstance(v, list):
        return [norm(x) for x in v]
    return v


def main():
    gold, ark = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
    bad = []
    for sec, vals in ark.items():
        for k, v in vals.items():
            if sec not in gold or k not in gold[sec]:
                print(f"  (extra in dump, not in golden: {sec}.{k})"); continue
            if norm(gold[sec][k]) != norm(v):
                bad.append(f"{sec}.{k}")
    if bad:
        print("MISMATCH against arkworks:", ", ".join(bad)); sys.exit(1)
    print("all", sum(len(v) for v in ark.values()), "known answers equal arkworks' -- oracle pinned")


if __name__ == "__main__":
    main()
