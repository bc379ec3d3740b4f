#!/bin/bash
# Same-call A/B of an environment switch on a bench workload: tools/probes/ab_env.sh VAR=VALUE [workload=c3] [steps=20]
cd "$(dirname "$0")/../.."
KV=$1; WL=${2:-c3}; ST=${3:-20}
one() { python bench.py --workload $WL --steps $ST --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo -n "default: "; one; echo -n "$KV: "; env $KV bash -c "$(declare -f one); WL=$WL ST=$ST one"; done
