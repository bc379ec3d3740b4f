#!/bin/bash
# In-run comparison of the default library with variants lib/libl4p_hip_<v>.so (make VARIANT=<v> ...):  tools/ab3.sh c3 "b c" [pattern]
cd "$(dirname "$0")/.."
WL=${1:-c3}; VS=${2:-b}; PAT=${3:-attention}
one() { python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo -n "A: "; one; for v in $VS; do echo -n "$v: "; L4P_HIP_LIB=$PWD/l4p_amd/lib/libl4p_hip_$v.so one; done; done
echo "--- per shape A"; python tools/prof_detail.py $WL 3 2>/dev/null | grep -E "$PAT" | head -40
for v in $VS; do echo "--- per shape $v"; L4P_HIP_LIB=$PWD/l4p_amd/lib/libl4p_hip_$v.so python tools/prof_detail.py $WL 3 2>/dev/null | grep -E "$PAT" | head -40; done
