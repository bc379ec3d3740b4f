#!/bin/bash
# Same-call A/B of the two-workgroups-per-CU GEMM form (csrc/gemm4w.hpp; L4P_GEMM_4W: 0 = off, 1 = shapes with >= 2 rounds of
# 256 x 128 tiles, 2 = every shape the 8-phase kernel would take): its parity tests, alternating c3 bench runs, per-shape profiles.
#   gpurun -- bash tools/probes/ab_4w.sh [out dir name]
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-ab4w}
mkdir -p $O
timeout 900 python -m pytest tests/test_gemm4w_gpu.py -x -q 2>&1 | tail -15 > $O/pytest.log
one() { python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do for v in 0 1 2; do echo -n "c3 4W=$v: "; L4P_GEMM_4W=$v one c3; done; done > $O/ab.txt 2>&1
for v in 0 1 2; do L4P_GEMM_4W=$v python tools/prof_detail.py c3 3 > $O/shapes_4w$v.txt 2>/dev/null; done
cat $O/pytest.log $O/ab.txt
python - $O <<'PY'
import re, sys
O = sys.argv[1]
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r'(\w+)\s+(M\d+ N\d+ K\d+ epi\d act\d) (.*?)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', l)
        if m: d[(m.group(1), m.group(2))] = (m.group(3), float(m.group(5)), float(m.group(6)))
    return d
a, b, c = (load(f'{O}/shapes_4w{v}.txt') for v in (0, 1, 2))
for k in sorted(a, key=lambda k: -a[k][1]):
    if k in c and (a[k][0] != c[k][0] or (k in b and a[k][0] != b[k][0])):
        print(f"{k[1]:34s} {a[k][0]:18s} {a[k][2]:8.1f} us | 4W=1 {b.get(k, ('', 0, 0))[2]:8.1f} | 4W=2 {c[k][0]:18s} {c[k][2]:8.1f} us ({c[k][2] / a[k][2]:.2f}x)")
PY
