#!/bin/bash
# the emulated rank of configs[4] with the decoders confined to a part of the chip (the tracker's small kernels then never wait for a
# round of conv workgroups to end)
for m in ${MASKS:-"" "0,224" "0,192" "0,160" "0,128"}; do
  echo "=== decoders on CUs [$m]"
  C5_TL_DEC_CUS=$m C5_TL_ORDER=dec_first python tools/probes/c5_rank_timeline.py 2>&1 | grep -i "segment A\|Error\|Traceback" | sort -u
done
