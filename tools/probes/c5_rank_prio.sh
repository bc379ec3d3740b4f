#!/bin/bash
# the emulated rank of configs[4]: tracker streams at default / high priority, no profiler attached
for p in 0 1; do
  echo "=== L4P_TRACK_PRIO=$p"
  L4P_TRACK_PRIO=$p C5_TL_ORDER=dec_first python tools/probes/c5_rank_timeline.py 2>&1 | grep -i "segment\|host\|Error\|Traceback" | sort -u
done
