#!/bin/bash
# In-bench A/B of the attention kernel forms at batch 4 (event-timed launch average inside the c3 step, not a standalone loop):
# default (query split, persistent), L4P_ATTN_PERSIST=0 (query split, one tile per workgroup), L4P_ATTN_VARIANT=2 (4-wave form)
cd "$(dirname "$0")/../.."
one() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['roofline_attention']; print('$1', d['value'], a['avg_launch_us'], a['frac'])"; }
for i in 1 2; do one X=0; one L4P_ATTN_PERSIST=0; one L4P_ATTN_VARIANT=2; done
