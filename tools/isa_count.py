"""Instruction mix of the kernels in a hipcc -S file: isa_count.py file.s [name-substring]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    vop3 = sum(v for k, v in c.items() if k.startswith(('v_mul_lo', 'v_mul_hi', 'v_mad_u64', 'v_alignbit', 'v_add3', 'v_lshl_add', 'v_lshl_or', 'v_and_or', 'v_xad', 'v_bfe', 'v_perm')))
    print(f"{name[:60]:60s} total {len(ins):6d} valu {valu:6d} vop3-ish {vop3:5d}  "
          + " ".join(f"{k}={v}" for k, v in c.most_common(12)))
