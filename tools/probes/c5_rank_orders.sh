#!/bin/bash
# the emulated rank of configs[4] with the three host issue orders, no profiler attached
for o in dec_first trk_first thread; do
  echo "=== order $o"
  C5_TL_ORDER=$o python tools/probes/c5_rank_timeline.py 2>&1 | grep -i "segment\|host\|Error\|Traceback" | sort -u
done
