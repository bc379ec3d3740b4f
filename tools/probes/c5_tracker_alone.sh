#!/bin/bash
# the tracker of one rank's query shard (8 of 64 queries, 31 windows) alone, with the knobs of this round's token-side kernels on / off
for k in ${KNOBS:-"L4P_GEMM_SKINNY"}; do
  for v in 0 1; do
    echo "=== $k=$v"
    env $k=$v C5_TL_ORDER=dec_first python tools/probes/c5_rank_timeline.py 2>&1 | grep -i "segment\|Error\|Traceback" | sort -u
  done
done
