#!/bin/bash
# same-call A/B of attention builds: default, _b, _c libraries (tools/attn_time.py, random data) + the kernel parity tests
cd "$(dirname "$0")/../.."
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention or qkv" 2>&1 | tail -3
for r in 1 2; do
for v in "" _b _c; do
  lib=$PWD/l4p_amd/lib/libl4p_hip$v.so
  [ -f $lib ] || continue
  echo "== lib '$v'"; L4P_HIP_LIB=$lib python tools/attn_time.py 2>/dev/null
done; done
