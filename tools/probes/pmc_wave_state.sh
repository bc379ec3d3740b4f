#!/bin/bash
# wave-state and instruction-mix counters of the c3 step's hot kernels (why the 8-phase GEMM keeps the matrix pipe at 44 % where the
# LDS-halo conv reaches 62 %): four PMC passes, tools/pmc_family_counters.py
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i + 1))
  rm -rf $O/pmc_ws_$i
  rocprofv3 --pmc $set --kernel-trace -d $O/pmc_ws_$i -o out -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>$O/pmc_ws_$i.err || echo "pass $i ($set) failed"
done
cd $R
python tools/pmc_family_counters.py $O/pmc_ws_1 $O/pmc_ws_2 $O/pmc_ws_3 $O/pmc_ws_4 > $O/r06_c3_wave_state_counters.md 2>$O/pmc_ws_report.err
rm -rf $O/pmc_ws_1 $O/pmc_ws_2 $O/pmc_ws_3 $O/pmc_ws_4; tail -14 $O/r06_c3_wave_state_counters.md; tail -3 $O/pmc_ws_report.err
