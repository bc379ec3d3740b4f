#!/bin/bash
# configs[4], one of eight ranks: the dense stitch after the tracker's join (default) / beside the tail of the recursion
for v in 0 1 0 1; do
  echo -n "L4P_TRACK_BESIDE_STITCH=$v: "
  L4P_TRACK_BESIDE_STITCH=$v python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one GPU', d['ms_per_step'], 'rank gather', d['emulated_rank0_of_8_ms'], d['implied_8gpu_speedup_emulated_rank'], 'seam', d['emulated_rank0_of_8_seam_local_ms'], d['implied_8gpu_speedup_emulated_rank_seam_local'])"
done
