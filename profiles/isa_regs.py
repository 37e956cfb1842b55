#!/usr/bin/env python3
"""Register / spill / LDS summary of every rollout kernel in a hipcc -S dump (AMDGPU metadata section)."""
import re, sys
s = open(sys.argv[1]).read()
md = s[s.index('amdhsa.kernels:'):]
for blk in md.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    g = lambda k: (re.search(r'\.' + k + r':\s+(\d+)', blk) or [None, '?'])[1]
    t = re.search(r'rollout_kernelILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)', name)
    label = ('rollout<F%s,K%s,forced%s,std%s,sig%s,ragged%s,gen%s>' % t.groups()) if t else name[:60]
    print(f"{label:58s} sgpr {g('sgpr_count'):>3} sspill {g('sgpr_spill_count'):>3} vgpr {g('vgpr_count'):>3} "
          f"vspill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}")
