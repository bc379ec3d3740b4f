#!/bin/bash
# later-window shared half of the layer-0 key projections (default) against every track on its own (L4P_TRACK_HALF_SHARE=0)
cd "$(dirname "$0")/../.."
one() { python bench.py --workload $1 --steps 3 --warmup 1 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for wl in demo c5; do for r in 1 2; do for v in 1 0; do echo -n "$wl HALF_SHARE=$v: "; L4P_TRACK_HALF_SHARE=$v one $wl; done; done; done
