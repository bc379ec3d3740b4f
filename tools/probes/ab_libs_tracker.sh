#!/bin/bash
# the 8-query tracker of configs[4] (and the emulated rank) on two builds of the library, alternating: L4P_HIP_LIB
cd "$(dirname "$0")/../.."
for i in 1 2; do
  for v in prev ""; do
    lib=$PWD/l4p_amd/lib/libl4p_hip${v:+_$v}.so
    echo "=== ${v:-shipped}"
    L4P_HIP_LIB=$lib C5_TL_ORDER=dec_first python tools/probes/c5_rank_timeline.py 2>&1 | grep -i "segment [AB]\|Error" | sort -u
  done
done
