#!/bin/bash
# per-shape times of the c3 step with the 8-phase kernel (default dispatch) and without it (L4P_GEMM_VARIANT=1)
cd "$(dirname "$0")/../.."
python tools/prof_detail.py c3 3 2>/dev/null | grep -E "^(gemm|conv3d)" | sort > /tmp/a.txt
L4P_GEMM_VARIANT=1 python tools/prof_detail.py c3 3 2>/dev/null | grep -E "^(gemm|conv3d)" | sort > /tmp/b.txt
python - <<'PY'
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r'(\w+)\s+(M\d+ N\d+ K\d+ epi\d act\d) (\S+ \S+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', l)
        if m: d[(m.group(1),m.group(2))]=(m.group(3),float(m.group(5)),float(m.group(6)))
    return d
a,b=load('/tmp/a.txt'),load('/tmp/b.txt')
for k in sorted(a, key=lambda k:-a[k][1]):
    if k in b and a[k][0]!=b[k][0]:
        print(f"{k[0]:7s} {k[1]:34s} {a[k][0]:14s} {a[k][2]:8.1f} us | {b[k][0]:14s} {b[k][2]:8.1f} us  ({b[k][2]/a[k][2]:.2f}x)")
PY
