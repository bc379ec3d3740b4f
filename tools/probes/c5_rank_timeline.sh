#!/bin/bash
# both enqueue orders of the emulated rank under a kernel trace (see c5_rank_timeline.py)
export TMPDIR=/tmp
for o in dec_first trk_first; do
  rm -rf /tmp/tl_$o
  C5_TL_ORDER=$o rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$o -- python tools/probes/c5_rank_timeline.py > gpurun_out/c5_tl_$o.log 2>gpurun_out/c5_tl_$o.err
  echo "=== order $o"; grep -A1 segment gpurun_out/c5_tl_$o.log
  python tools/probes/c5_rank_timeline_report.py /tmp/tl_$o
done
