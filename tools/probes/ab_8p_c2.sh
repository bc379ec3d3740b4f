#!/bin/bash
# per-shape times of the c2 (batch 1) step with the default dispatch and with the 8-phase kernel forced wherever it can run
# (L4P_GEMM_VARIANT=10)
cd "$(dirname "$0")/../.."
python tools/prof_detail.py c2 5 2>/dev/null | grep -E "^(gemm|conv3d)" | sort > /tmp/a.txt
L4P_GEMM_VARIANT=10 python tools/prof_detail.py c2 5 2>/dev/null | grep -E "^(gemm|conv3d)" | sort > /tmp/b.txt
python - <<'PY'
import re
def load(p):
    d={}
    for l in open(p):
        m=re.match(r'(\w+)\s+(M\d+ N\d+ K\d+ epi\d act\d) (\S+ \S+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', l)
        if m: d[(m.group(1),m.group(2))]=(m.group(3),float(m.group(5)),float(m.group(6)),float(m.group(4)))
    return d
a,b=load('/tmp/a.txt'),load('/tmp/b.txt')
for k in sorted(a, key=lambda k:-a[k][1]):
    if k in b and a[k][0]!=b[k][0]:
        print(f"{k[0]:7s} {k[1]:34s} x{a[k][3]:4.0f} {a[k][0]:14s} {a[k][2]:8.1f} us | {b[k][0]:14s} {b[k][2]:8.1f} us  ({b[k][2]/a[k][2]:.2f}x)")
PY
