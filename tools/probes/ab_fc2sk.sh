#!/bin/bash
# the batch-1 MLP-out projection: 5 slices on the 8-phase kernel (default) against 2 slices on 128x128 tiles (L4P_FC2_SPLITK8=0)
cd "$(dirname "$0")/../.."
one() { python bench.py --workload c2 --steps 40 --warmup 10 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do for v in 1 0; do echo -n "c2 FC2_SPLITK8=$v: "; L4P_FC2_SPLITK8=$v one; done; done
for v in 1 0; do echo "--- per shape FC2_SPLITK8=$v"; L4P_FC2_SPLITK8=$v python tools/prof_detail.py c2 5 2>/dev/null | grep -E "K6144|total"; done
