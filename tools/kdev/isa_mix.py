#!/usr/bin/env python3
"""Instruction mix per kernel of the device ISA (hipcc -S --cuda-device-only): python3 tools/kdev/isa_mix.py /tmp/isa/engine.s [name-substring ...]"""
import re, sys, collections
s = open(sys.argv[1]).read()
want = sys.argv[2:]
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S):
    name, body = m.group(1), m.group(2)
    if want and not any(k in name for k in want): continue
    c = collections.Counter()
    for l in body.split('\n'):
        l = l.strip()
        if not l or l[0] in '.;/' or l.endswith(':'): continue
        c[l.split()[0]] += 1
    tot = sum(c.values())
    scr = sum(v for k, v in c.items() if k.startswith('scratch_'))
    print("%-60s total %6d mad %5d addc %5d s_nop %5d scratch %4d mov %5d ds %4d waitcnt %4d" % (name[:60], tot, c['v_mad_u64_u32'], c['v_addc_co_u32'], c['s_nop'], scr,
          c['v_mov_b32'] + c['v_accvgpr_write_b32'] + c['v_accvgpr_read_b32'], sum(v for k, v in c.items() if k.startswith('ds_')), c['s_waitcnt']))
