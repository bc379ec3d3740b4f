#!/bin/bash
# After a kernel-source edit that leaves the hot kernels alone: re-take what is tied to the kernel tree hash (HBM bytes per launch of the
# c3 step, the c3 line that carries them) and the lines the edit moved (c5, demo), without the full tools/make_profiles.sh.
#   tools/refresh_hash_tied.sh r06      (through gpurun, from the repo root; copy gpurun_out/<tag>_* into profiles/ afterwards)
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_${TAG}_fetch $O/pmc_${TAG}_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
cd $R
python tools/pmc_hbm_traffic.py $O/pmc_${TAG}_fetch $O/pmc_${TAG}_write $O/${TAG}_c3_hbm_traffic > /dev/null
cp $O/${TAG}_c3_hbm_traffic.json $O/${TAG}_c3_hbm_traffic.md $R/profiles/
python bench.py --steps 20 --warmup 5 > $O/${TAG}_c3_bench_line.json 2> $O/${TAG}_c3.err
python bench.py --workload c5 --steps 3 --warmup 1 > $O/${TAG}_c5_bench_line.json 2> $O/${TAG}_c5.err
python bench.py --workload demo --steps 3 --warmup 1 > $O/${TAG}_demo_bench_line.json 2> $O/${TAG}_demo.err
python tools/prof_detail.py c3 5 > $O/${TAG}_c3_per_shape_event_profile.txt 2>/dev/null
python tools/prof_detail.py c5 2 8 > $O/${TAG}_c5_rank_shard_per_shape_event_profile.txt 2>/dev/null
KNOBS="L4P_GEMM_SKINNY L4P_TRACK_DEEP L4P_TRACK_FOLD_L0 L4P_TRACK_KWIN L4P_READOUT_WIDE" bash tools/probes/c5_tracker_alone.sh > $O/${TAG}_c5_rank_shard_knobs.txt 2>&1
ls -la $O | grep ${TAG}_ | head -20
