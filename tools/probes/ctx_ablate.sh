#!/bin/bash
# Build ablated variants of the context kernel (track.hip -DCTX_ABL=n; wrong results, timing only) and time them with ctx_time.py.
# On the build host:  bash tools/probes/ctx_ablate.sh build     On the GPU box:  bash tools/probes/ctx_ablate.sh
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  for n in 1 2 3; do
    rm -rf build/obj_ctxabl$n; cp -r build/obj build/obj_ctxabl$n; rm -f build/obj_ctxabl$n/track.o
    make -C l4p_amd/csrc VARIANT=ctxabl$n EXTRA=-DCTX_ABL=$n ISA_LINT=0 2>&1 | grep -i "error" 
  done
  exit 0
fi
echo "=== shipped"; python tools/probes/ctx_time.py 2>&1 | grep "N="
for n in 1 2 3; do
  echo "=== CTX_ABL=$n (1: no LDS reads / MFMAs; 2: no stage requests inside the loop; 3: no barrier)"
  L4P_HIP_LIB=$PWD/l4p_amd/lib/libl4p_hip_ctxabl$n.so python tools/probes/ctx_time.py 2>&1 | grep "N="
done
