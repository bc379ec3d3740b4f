#!/bin/bash
# Same-call A/B of the 256x192 tile form of the 8-phase GEMM (L4P_GEMM_T192=1, default) against 256x256 tiles only (=0): the
# 8-phase parity tests, two alternating c3 bench runs each, and the per-shape profile of both (gpurun -- bash tools/probes/ab_t192.sh)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3h
timeout 900 python -m pytest tests/test_gemm8p_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/r3h/pytest.log
one() { python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo -n "c3 T192=1: "; one c3; echo -n "c3 T192=0: "; L4P_GEMM_T192=0 one c3; done > gpurun_out/r3h/ab.txt 2>&1
python tools/prof_detail.py c3 3 > gpurun_out/r3h/shapes_t192.txt 2>/dev/null
L4P_GEMM_T192=0 python tools/prof_detail.py c3 3 > gpurun_out/r3h/shapes_t256.txt 2>/dev/null
cat gpurun_out/r3h/pytest.log gpurun_out/r3h/ab.txt
