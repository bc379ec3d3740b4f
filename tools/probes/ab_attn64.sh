#!/bin/bash
# same-call A/B of the two encoder attention kernels (knob attn64: 0 = 8 waves x 32 rows, 1 = 4 waves x 64 rows) + parity tests
cd "$(dirname "$0")/../.."
python -m pytest tests/test_kernels_gpu.py tests/test_gemm8p_gpu.py -m gpu -q -x -k "attention or qkv" 2>&1 | tail -5
for r in 1 2; do
for v in 0 1; do
  echo "== attn64=$v"; L4P_ATTN64=$v python tools/attn_time.py 2>&1 | tail -4
done; done
