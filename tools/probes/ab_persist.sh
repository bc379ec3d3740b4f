#!/bin/bash
# Same-call A/B of the persistent form of the 8-phase GEMM (L4P_GEMM_PERSIST=1, default) against the plain launch (=0)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -5
one() { python bench.py --workload $1 --steps $2 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo -n "c3 persist=1: "; one c3 20; echo -n "c3 persist=0: "; L4P_GEMM_PERSIST=0 one c3 20; done
echo -n "demo persist=1: "; one demo 3; echo -n "demo persist=0: "; L4P_GEMM_PERSIST=0 one demo 3
python tools/prof_detail.py c3 3 > gpurun_out/r3j/shapes_p1.txt 2>/dev/null
L4P_GEMM_PERSIST=0 python tools/prof_detail.py c3 3 > gpurun_out/r3j/shapes_p0.txt 2>/dev/null
grep -E "^gemm" gpurun_out/r3j/shapes_p1.txt | head -14; echo; grep -E "^gemm" gpurun_out/r3j/shapes_p0.txt | head -14
