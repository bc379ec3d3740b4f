#!/bin/bash
# In-run A/B of two builds of libl4p_hip (boxes of the pool differ by +-4 %, so only same-call comparisons mean anything):
#   make -C l4p_amd/csrc VARIANT=b EXTRA=-DSOMETHING ; gpurun -- tools/ab.sh [workload=c3] [pattern for the per-shape lines]
# Runs the bench alternately with the default library (A) and lib/libl4p_hip_b.so (B), then the per-shape profile of both.
cd "$(dirname "$0")/.."
WL=${1:-c3}; PAT=${2:-total}
B=$PWD/l4p_amd/lib/libl4p_hip_b.so
one() { python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo -n "A: "; one; echo -n "B: "; L4P_HIP_LIB=$B one; done
echo "--- per shape A"; python tools/prof_detail.py $WL 3 2>/dev/null | grep -E "$PAT"
echo "--- per shape B"; L4P_HIP_LIB=$B python tools/prof_detail.py $WL 3 2>/dev/null | grep -E "$PAT"
