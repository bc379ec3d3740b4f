# A/B of the up-sampling kernel's items in flight per thread (L4P_UPS_IPT): bash tools/probes/ab_ups.sh
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm8p_gpu.py tests/test_encoder_dpt_gpu.py -q -x -k "upsampl or trilinear or interp or dpt" 2>&1 | tail -3
for r in 1 2; do
for v in 1 2 4; do
L4P_UPS_IPT=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_ups_$v.json 2>/dev/null
python - <<PY
import json
r=json.loads(open("gpurun_out/ab_ups_$v.json").read().strip().splitlines()[-1])
print("c3 ups_ipt=$v", r["value"], "frames/s  elementwise ms/step", r["kernel_classes"]["elementwise"]["ms_per_step"])
PY
done
done
