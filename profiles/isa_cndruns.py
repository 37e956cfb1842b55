#!/usr/bin/env python3
"""Runs of consecutive VOP2-encoded v_cndmask_b32 (implicit VCC mask) per basic block of one kernel in a hipcc -S dump.
profiles/micro/cnd_rates.hip: on gfx950 three or more of them back to back cost 16-19 cycles EACH (5 for one or two,
5 for the _e64 encoding whatever the mask register).  usage: isa_cndruns.py file.s kernel-substring"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = end = None
for i, l in enumerate(s):
    if l.startswith('_ZN') and pat in l.split(':')[0] and l.split(';')[0].strip().endswith(':'): start = i
    if start is not None and l.strip().startswith('.amdhsa_kernel'): end = i; break
blk = 'entry'; run = 0; tot = {}
def flush():
    global run
    if run >= 3: tot.setdefault(blk, []).append(run)
    run = 0
for l in s[start + 1:end]:
    t = l.strip()
    m = re.match(r'(\.LBB\d+_\d+):', t)
    if m: flush(); blk = m.group(1); continue
    if not t or t.startswith((';', '.', '//')): continue
    op = t.split()[0]
    if op == 'v_cndmask_b32_e32': run += 1
    elif op in ('s_nop',): pass
    else: flush()
flush()
for b, r in tot.items(): print(b, r, 'sum', sum(r))
print('total e32 selects in runs of >= 3:', sum(sum(r) for r in tot.values()))
