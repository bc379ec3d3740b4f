#!/bin/bash
# The 16-bit engines' drift gates, with their ratios printed, under the default kernels and with the round-5 / round-6 kernels whose
# arithmetic is NOT bit-identical to their predecessors switched off one by one (round-5 advisor finding: were the gates loosened to
# make room for them?):   tools/drift_report.sh > profiles/rNN_bf16_drift_ratios.txt      (through gpurun, from the repo root)
#   ratio = engine drift against the reference's fp32 golden / the reference's OWN autocast drift on the same inputs (rel-L2);
#   gate: every key <= 1.25 (2.0 for outputs of fewer than 4096 values), geometric mean <= 1.
cd "$(dirname "$0")/.."
SEL='tests/test_full_model_gpu.py tests/test_track_gpu.py tests/test_encoder_dpt_gpu.py tests/test_stitch_affine_gpu.py tests/test_long_recursion_gpu.py'
KEY='bf16 or 16-mixed or benchmarked or batch4 or batch8'
run() {
  echo "=== $1"
  env $2 python -m pytest $SEL -m gpu -q -s -k "$KEY" 2>&1 | grep -E "drift / reference autocast drift|tracks differ|passed|failed" | sed 's/^/  /'
}
run "defaults (ln_rows16 = 1, maskdot_mfma = 1, attn64 = 1)" "L4P_NOP=1"
run "ln_rows16 = 0 (one wave per LayerNorm3d row: the round-4 summation order)" "L4P_LN_ROWS16=0"
run "maskdot_mfma = 0 (all-VALU mask product: float row, float dot products)" "L4P_MASKDOT_MFMA=0"
run "attn64 = 0 (8-wave attention, rescale decided per 32 rows)" "L4P_ATTN64=0"
run "L4P_TRACK_FOLD_L0 = 0 (later windows: layer 0's token -> image attention on projected keys / values, the form up to round 6)" "L4P_TRACK_FOLD_L0=0"
