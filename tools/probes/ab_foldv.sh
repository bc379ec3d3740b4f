#!/bin/bash
# A/B of the folded value projection of the token -> image attentions (L4P_TRACK_FOLD_T2I_V) on one box: unit + tracker tests,
# then c3 alternately with the switch off / on, the demo case, and the per-shape profile.   usage: ab_foldv.sh <outdir>
O=gpurun_out/${1:-foldv}
mkdir -p $O
(timeout 1500 python -m pytest tests/test_track_gpu.py -x -q 2>&1 | tail -15) > $O/pytest.log
for rep in 1 2; do
  for v in 0 1; do
    L4P_TRACK_FOLD_T2I_V=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c3 FOLD_V=$v:', d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
for v in 0 1; do
  L4P_TRACK_FOLD_T2I_V=$v python bench.py --workload demo --steps 3 --warmup 1 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('demo FOLD_V=$v:', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
python tools/prof_detail.py c3 5 > $O/c3_shapes.txt 2>/dev/null
cat $O/pytest.log $O/ab.txt; head -40 $O/c3_shapes.txt
