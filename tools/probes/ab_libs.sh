#!/bin/bash
# same-call comparison of library builds (default, _b, _c ...): bench value + selected per-shape lines
#   tools/probes/ab_libs.sh <workload> <egrep pattern for per-shape lines> [suffixes...]
cd "$(dirname "$0")/../.."
WL=${1:-c3}; PAT=${2:-total}; shift 2; SFX=${@:-"- _b"}
one() { python bench.py --workload $WL --steps 15 --warmup 4 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do for s in $SFX; do [ "$s" = "-" ] && s=""; echo -n "lib '$s': "; L4P_HIP_LIB=$PWD/l4p_amd/lib/libl4p_hip$s.so one; done; done
for s in $SFX; do [ "$s" = "-" ] && s=""; echo "--- per shape, lib '$s'"; L4P_HIP_LIB=$PWD/l4p_amd/lib/libl4p_hip$s.so python tools/prof_detail.py $WL 3 2>/dev/null | grep -E "$PAT"; done
