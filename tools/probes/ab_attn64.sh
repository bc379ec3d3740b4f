#!/bin/bash
# the 64-row AGPR attention kernel (L4P_ATTN64=1) against the 32-row kernel, and builds of it with other read-ahead depths
cd "$(dirname "$0")/../.."
for r in 1 2; do
echo "== 32-row"; L4P_ATTN64=0 timeout 300 python tools/attn_time.py 2>/dev/null | head -1
for v in "" _b _c _d; do [ -f l4p_amd/lib/libl4p_hip$v.so ] || continue; echo "== 64-row lib '$v'"; L4P_HIP_LIB=$PWD/l4p_amd/lib/libl4p_hip$v.so L4P_ATTN64=1 timeout 300 python tools/attn_time.py 2>/dev/null | head -1; done; done
