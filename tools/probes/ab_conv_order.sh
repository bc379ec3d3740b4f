#!/bin/bash
# conv3_halo block order (t fastest vs round 3's w fastest = lib _b built with -DCONV_HALO_ORDER_WTH): step time, per-shape conv time,
# and HBM bytes fetched per launch (rocprofv3 --pmc FETCH_SIZE, x2 per the gfx950 correction)
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
tools/ab.sh c3 "halo"
cd /tmp && export TMPDIR=/tmp
for v in "" _b; do
  rm -rf /tmp/pf$v /tmp/pw$v
  L4P_HIP_LIB=$R/l4p_amd/lib/libl4p_hip$v.so rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf$v -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
  L4P_HIP_LIB=$R/l4p_amd/lib/libl4p_hip$v.so rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw$v -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
  echo "== lib '$v'"; python $R/tools/pmc_hbm_traffic.py /tmp/pf$v /tmp/pw$v /tmp/traffic$v > /dev/null; grep -E "conv3_halo|^\| class|conv3d" /tmp/traffic$v.md | cut -c1-200
done
