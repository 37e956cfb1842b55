#!/usr/bin/env python3
"""Per-basic-block instruction mix (all instruction kinds: the wavefront issues one instruction of any kind per
4 cycles) of one kernel in a hipcc -S dump.  usage: isa_mix.py file.s kernel-substring [min_block_size]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = end = None
for i, l in enumerate(s):
    if l.startswith('_ZN') and pat in l.split(':')[0] and l.split(';')[0].strip().endswith(':'): start = i
    if start is not None and l.strip().startswith('.amdhsa_kernel'): end = i; break
blocks = []; cur = ['entry', []]
for l in s[start + 1:end]:
    t = l.strip()
    m = re.match(r'(\.LBB\d+_\d+):', t)
    if m: blocks.append(cur); cur = [m.group(1), []]; continue
    if not t or t.startswith((';', '.', '//')): continue
    cur[1].append(t)
blocks.append(cur)
for name, ins in blocks:
    ops = [x.split()[0] for x in ins]
    c = lambda p: sum(1 for o in ops if re.match(p, o))
    if len(ops) >= mn:
        print(f"{name:11s} n={len(ops):4d} valu={c(r'v_'):4d} salu={c(r's_(?!waitcnt|nop|cbranch|branch|load)'):4d} smov={c(r's_mov'):3d} "
              f"vmov={c(r'v_mov'):3d} sload={c(r's_load'):2d} wait={c(r's_waitcnt|s_nop'):2d} br={c(r's_c?branch'):2d} ds={c(r'ds_'):2d} gl={c(r'global_'):2d} "
              f"lane={c(r'v_(read|write)lane'):2d} cnd={c(r'v_cndmask'):3d} f64={c(r'v_.*f64'):3d}")
