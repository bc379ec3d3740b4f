#!/bin/bash
# One of eight ranks of configs[4] under a kernel trace (c5_rank_timeline.py): the decoders on the whole chip, and on the CUs
# parallel.decoder_stream confines them to beside a small query shard.  Per segment: span, busy time per hardware queue, overlap.
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for m in all default; do
  rm -rf /tmp/tl_$m
  if [ $m = all ]; then export C5_TL_DEC_CUS=""; else unset C5_TL_DEC_CUS; fi
  C5_TL_ORDER=dec_first rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$m -- python tools/probes/c5_rank_timeline.py > /tmp/tl_$m.log 2>/tmp/tl_$m.err
  echo "=== decoders on: $m CUs (parallel.decoder_stream)"; grep -h "segment\|host" /tmp/tl_$m.err | sort -u
  python tools/probes/c5_rank_timeline_report.py /tmp/tl_$m
done
