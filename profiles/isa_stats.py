#!/usr/bin/env python3
"""Counts instruction classes per kernel in a hipcc -S dump (helper for DESIGN.md / tuning)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else 'rollout_kernelILi1ELi3ELb1'
start = None
for i, l in enumerate(s):
    if l.startswith('_ZN') and l.split(':')[0].find(pat) >= 0 and l.rstrip().split(';')[0].strip().endswith(':'):
        start = i
    if start is not None and l.strip().startswith('.amdhsa_kernel'):
        end = i; break
body = s[start:end]
ins = []
for l in body:
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')) or t.split(';')[0].strip().endswith(':'):
        continue
    ins.append(t.split()[0])
c = Counter(ins)
def grp(p): return sum(v for k, v in c.items() if re.match(p, k))
print('kernel', pat, 'total', len(ins))
print('f64 valu', grp(r'v_.*_f64'), '| f32 valu', grp(r'v_.*_f32'), '| scratch', grp(r'scratch_'), '| s_load', grp(r's_load'),
      '| int mul', grp(r'v_mul_(hi|lo)|v_mad_u64'), '| ds', grp(r'ds_'), '| global', grp(r'global_'), '| flat', grp(r'flat_'),
      '| lane xfer', grp(r'v_(read|write)lane'), '| branches', grp(r's_cbranch'), '| v_div/rcp/sqrt', grp(r'v_(div|rcp|sqrt|rsq)'))
print(c.most_common(45))
