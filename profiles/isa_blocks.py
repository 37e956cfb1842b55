#!/usr/bin/env python3
"""Per-basic-block instruction summary of one kernel in a hipcc -S dump."""
import re, sys
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = end = None
for i, l in enumerate(s):
    if l.startswith('_ZN') and pat in l.split(':')[0] and l.split(';')[0].strip().endswith(':'):
        start = i
    if start is not None and l.strip().startswith('.amdhsa_kernel'):
        end = i; break
blocks = []; cur = ['entry', []]
for l in s[start+1:end]:
    t = l.strip()
    if not t or t.startswith(('.', ';', '//')):
        if t.startswith('.LBB') and t.split(';')[0].strip().endswith(':'):
            blocks.append(cur); cur = [t.split(':')[0], []]
        continue
    if t.split(';')[0].strip().endswith(':'):
        blocks.append(cur); cur = [t.split(':')[0], []]; continue
    cur[1].append(t)
blocks.append(cur)
tot = 0
for name, ins in blocks:
    ops = [x.split()[0] for x in ins]
    n = len(ops)
    def c(p): return sum(1 for o in ops if re.match(p, o))
    br = [x for x in ins if x.startswith(('s_cbranch', 's_branch'))]
    tgt = ','.join(b.split()[-1] for b in br)
    print(f"{name:12s} n={n:4d} f64={c(r'v_.*_f64'):3d} f32={c(r'v_.*_f32'):3d} mad64={c(r'v_mad_u64'):3d} scr={c(r'scratch_'):3d} lane={c(r'v_(read|write)lane'):3d} ds={c(r'ds_'):2d} gl={c(r'global_'):2d} -> {tgt}")
