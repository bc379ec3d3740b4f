#!/bin/bash
# configs[4] on one GPU, one of eight ranks emulated (bench.py --workload c5: emulated_rank0_of_8_ms), with CU-masked streams for the
# decoders / the tracker of the rank (L4P_C5_DEC_CUS / L4P_C5_TRK_CUS = "first,count").   usage: ab_c5_masks.sh <outdir>
O=gpurun_out/${1:-c5masks}
mkdir -p $O
run() { env $2 $3 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'rank0/8:', d.get('emulated_rank0_of_8_ms'), 'x', d.get('implied_8gpu_speedup_emulated_rank'), 'trk8', d.get('phase3_track_ms_on_an_eighth_of_the_queries'), 'dec', d.get('phase1b_decoders_ms'))" >> $O/ab.txt; }
run "none              " A=1 B=1
run "dec 32..256       " L4P_C5_DEC_CUS=32,224 B=1
run "dec 32.. trk 0..32" L4P_C5_DEC_CUS=32,224 L4P_C5_TRK_CUS=0,32
run "dec 64.. trk 0..64" L4P_C5_DEC_CUS=64,192 L4P_C5_TRK_CUS=0,64
run "dec 64..256       " L4P_C5_DEC_CUS=64,192 B=1
run "dec 128.. trk ..128" L4P_C5_DEC_CUS=128,128 L4P_C5_TRK_CUS=0,128
cat $O/ab.txt; tail -3 $O/err.txt
