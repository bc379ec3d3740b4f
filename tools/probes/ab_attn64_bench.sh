#!/bin/bash
# same-call A/B of the c3 bench with the attn64 knob off / on, then the whole GPU suite with its slowest tests listed
cd "$(dirname "$0")/../.."
for r in 1 2; do for v in 0 1; do echo "== attn64=$v"; L4P_ATTN64=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'attn', d['roofline_attention']['frac'], d['roofline_attention'].get('avg_launch_us'), 'gemm', d['roofline_gemm']['frac'], 'conv', d['roofline']['frac'])"; done; done
if [ "$1" = "suite" ]; then python -m pytest tests -m gpu -x -q --durations=25 2>&1 | tail -45; fi
