#!/bin/bash
# Regenerate the round's measurement artefacts on a GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh r02
# Writes gpurun_out/<tag>_*: bench lines (c3, c2, c5, demo, prep), rocprofv3 --kernel-trace --stats summaries of the c3 / c2
# commands, the per-shape event profile, and HBM bytes per launch from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE;
# --kernel-trace only, as gpurun requires), matrix-pipe utilisation + effective clock per hot kernel (SQ_VALU_MFMA_BUSY_CYCLES /
# GRBM_GUI_ACTIVE passes, tools/pmc_mfma_util.py) with the rocm-smi power trace.  Copy what should be judged into profiles/.
TAG=${1:-r02}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
cd $R
cd /tmp && export TMPDIR=/tmp
# HBM bytes per launch first: bench.py attaches them to its roofline objects only from a file taken on THESE kernels (kernel tree hash)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
cd $R
python tools/pmc_hbm_traffic.py $O/pmc_${TAG}_fetch $O/pmc_${TAG}_write $O/${TAG}_c3_hbm_traffic > /dev/null
cp $O/${TAG}_c3_hbm_traffic.json $O/${TAG}_c3_hbm_traffic.md $R/profiles/   # (on the box; copy them into the tracked profiles/ afterwards)
python bench.py --steps 20 --warmup 5 > $O/${TAG}_c3_bench_line.json 2> $O/${TAG}_c3.err
python bench.py --workload c2 --steps 40 --warmup 10 > $O/${TAG}_c2_bench_line.json 2> $O/${TAG}_c2.err
python bench.py --workload c5 --steps 3 --warmup 1 > $O/${TAG}_c5_bench_line.json 2> $O/${TAG}_c5.err
python bench.py --workload demo --steps 3 --warmup 1 > $O/${TAG}_demo_bench_line.json 2> $O/${TAG}_demo.err
python bench.py --workload prep --steps 50 --warmup 10 > $O/${TAG}_prep_bench_line.json 2> $O/${TAG}_prep.err
python bench.py --batch 8 --steps 10 --warmup 3 > $O/${TAG}_c3_batch8_bench_line.json 2> $O/${TAG}_c3_batch8.err
# the other two engines on the same workload: IEEE half (the reference demo's own "16-mixed") and the exact-f32 parity engine (priced
# against the 157 TF f32-input MFMA peak)
python bench.py --precision 16-mixed --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_c3_f16_bench_line.json 2> $O/${TAG}_c3_f16.err
python bench.py --precision 32-true --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_c3_f32_bench_line.json 2> $O/${TAG}_c3_f32.err
python tools/prof_detail.py c3 5 > $O/${TAG}_c3_per_shape_event_profile.txt 2>/dev/null
python tools/prof_detail.py c2 10 > $O/${TAG}_c2_per_shape_event_profile.txt 2>/dev/null
cd /tmp
CMD3="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof"
CMD2="python $R/bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline --no-prof"
# (tracker clips serialised on one stream, as in bench.py's own event-timed pass: with the clips on their own streams and the
#  dense decoders beside them, concurrent kernels share the chip and every one of them reports the shared interval)
L4P_TRACK_STREAMS=0 L4P_HEAD_STREAMS=0 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_c3 -o out -- $CMD3 > $O/prof_${TAG}_c3.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_c2 -o out -- $CMD2 > $O/prof_${TAG}_c2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_${TAG}_sq -o out -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_${TAG}_grbm -o out -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
cd $R
bash tools/probes/power_probe.sh > $O/${TAG}_power_probe.txt 2>&1
# FETCH_SIZE / WRITE_SIZE calibration by access width (the x2 of FETCH_SIZE is measured here, not assumed)
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_calib_f -- $R/tools/probes/fetch_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_calib_w -- $R/tools/probes/fetch_calib > /dev/null 2>&1
cd $R
python tools/pmc_fetch_calibration.py $O/pmc_${TAG}_calib_f $O/pmc_${TAG}_calib_w > $O/${TAG}_fetch_size_calibration.md 2>/dev/null
python tools/pmc_mfma_util.py $O/pmc_${TAG}_sq $O/pmc_${TAG}_grbm $O/${TAG}_power_probe.txt > $O/${TAG}_c3_mfma_util.md 2> $O/${TAG}_mfma_util.err
python tools/rocprof_summary.py $(find $O/prof_${TAG}_c3 -name "*.db" | head -1) "L4P_TRACK_STREAMS=0 L4P_HEAD_STREAMS=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof (c3: 7 steps, every kernel serialised on one stream)" > $O/${TAG}_c3_kernel_stats.md
python tools/rocprof_summary.py $(find $O/prof_${TAG}_c2 -name "*.db" | head -1) "python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline --no-prof (c2: 13 steps)" > $O/${TAG}_c2_kernel_stats.md
# round 6: the two encoder attention kernels side by side (knob attn64), random and all-zero operands; what fits behind an MFMA when
# one wave owns the SIMD (tools/probes/gap_probe.hip); the 16-bit drift gates' ratios with the non-bit-identical kernels off one by one
( for v in 0 1; do echo "== attn64=$v, random operands"; L4P_ATTN64=$v python tools/attn_time.py 2>/dev/null | tail -3; echo "== attn64=$v, all-zero operands"; L4P_ATTN64=$v python tools/attn_time.py --zeros 2>/dev/null | tail -3; done ) > $O/${TAG}_attention64_ab.txt 2>&1
[ -x tools/probes/gap_probe ] && timeout 300 tools/probes/gap_probe > $O/${TAG}_mfma_gap_probe.txt 2>&1
[ -x tools/probes/attn64_var_trace ] && ( timeout 60 tools/probes/attn64_var_trace 4; timeout 60 tools/probes/attn64_var_trace 4 1 ) > $O/${TAG}_attention64_step_trace.txt 2>&1
bash tools/drift_report.sh > $O/${TAG}_bf16_drift_ratios.txt 2>&1
# round 6, one of eight ranks of configs[4]: its query shard's tracker kernel by kernel, with this round's token-side forms on / off, the
# rank's timeline (decoders beside the tracker) and the decoders' CU mask
python tools/prof_detail.py c5 2 8 > $O/${TAG}_c5_rank_shard_per_shape_event_profile.txt 2>/dev/null
KNOBS="L4P_GEMM_SKINNY L4P_TRACK_DEEP L4P_TRACK_FOLD_L0 L4P_TRACK_KWIN L4P_READOUT_WIDE" bash tools/probes/c5_tracker_alone.sh > $O/${TAG}_c5_rank_shard_knobs.txt 2>&1
bash tools/probes/c5_rank_masks.sh > $O/${TAG}_c5_rank_decoder_cu_masks.txt 2>&1
bash tools/probes/c5_rank_timeline.sh > $O/${TAG}_c5_rank_timeline.txt 2>&1
ls -la $O | grep ${TAG}_ | head -40
