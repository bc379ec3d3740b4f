#!/bin/bash
# Same-call A/B: the attention kernel as shipped (A) against the -DATTN_DBG_MFMA16 build (B: every 32x32x16 MFMA issued as two
# 16x16x32 on the same operands - wrong results, same matrix-pipe cycles and FLOPs), alone on the encoder shape.
#   make -C l4p_amd/csrc VARIANT=b EXTRA=-DATTN_DBG_MFMA16 ; gpurun -- bash tools/probes/ab_attn16.sh
cd "$(dirname "$0")/../.."
B=$PWD/l4p_amd/lib/libl4p_hip_b.so
for i in 1 2; do
  echo "A (32x32x16):"; python tools/attn_time.py 2>/dev/null
  echo "B (2 x 16x16x32):"; L4P_HIP_LIB=$B python tools/attn_time.py 2>/dev/null
done
