#!/bin/bash
# Same-call A/B of an environment switch: gpurun -- bash tools/probes/ab_env.sh <workload> <VAR> <value A> <value B> [extra bench args]
cd "$(dirname "$0")/../.."
WL=$1; VAR=$2; A=$3; B=$4; shift 4
one() { env $VAR=$1 python bench.py --workload $WL --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-prof "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('emulated_rank0_of_8_ms',''))"; }
for i in 1 2 3; do echo -n "$VAR=$A: "; one $A "$@"; echo -n "$VAR=$B: "; one $B "$@"; done
