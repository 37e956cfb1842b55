#!/usr/bin/env python3
"""Opcode histogram of one kernel in a hipcc -S dump, with a rough issue-cost weight per opcode class
(wave64 on a 16-lane SIMD: 4 cycles for full-rate VALU, 16 for quarter-rate transcendental / 32-bit integer
multiply, 8/16 for fp64 per MI355X_MICROARCH.md's rate table where it names them).  Static counts: every
instruction of the kernel body once, so rare-path code is counted as heavily as the step loop — use with
isa_mix.py's per-block view.  usage: isa_ophist.py file.s kernel-substring [top]"""
import collections, re, sys
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]; top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = end = None
for i, l in enumerate(s):
    if l.startswith('_ZN') and pat in l.split(':')[0] and l.split(';')[0].strip().endswith(':'): start = i
    if start is not None and l.strip().startswith('.amdhsa_kernel'): end = i; break
h = collections.Counter()
for l in s[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((';', '.', '//')) or t.endswith(':'): continue
    h[t.split()[0]] += 1
tot = sum(h.values())
print('instructions', tot, 'valu', sum(v for k, v in h.items() if k.startswith('v_')))
for k, v in h.most_common(top): print(f'{k:28s} {v:6d}')
quarter = sum(v for k, v in h.items() if re.match(r'v_(mul_lo|mul_hi|mad_u64|mad_i64|rcp|rsq|sqrt|sin|cos|exp|log)', k))
print('quarter-rate class (int mul, transcendental):', quarter)
