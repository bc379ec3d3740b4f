# A/B of the LayerNorm knobs inside the bench: bash tools/probes/ab_ln.sh
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_track_gpu.py -q -x -k "layernorm or token_ordered" 2>&1 | tail -3
for r in 1 2 3; do
for v in 0 1; do
L4P_LN_ROWS16=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_ln_$v.json 2>/dev/null
python - <<PY
import json
r=json.loads(open("gpurun_out/ab_ln_$v.json").read().strip().splitlines()[-1])
print("c3 ln_rows16=$v", r["value"], "frames/s  LN ms/step", r["kernel_classes"]["layernorm"]["ms_per_step"])
PY
done
done
for v in 0 1; do L4P_LN_ROWS16=$v python tools/prof_detail.py c3 5 2>/dev/null | grep "layernorm  *M1048576"; done
python tools/prof_detail.py c5 2 2>/dev/null | grep "layernorm"
