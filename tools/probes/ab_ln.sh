set -e
timeout 600 python -m pytest tests/test_track_gpu.py -q -x -k "token_ordered or layernorm_chain" 2>&1 | tail -3
for r in 1 2 3; do
for v in 0 1; do
L4P_LN_TRACKS=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_ln_$v.json 2>/dev/null
python - <<PY
import json
r=json.loads(open("gpurun_out/ab_ln_$v.json").read().strip().splitlines()[-1])
k=r["kernel_classes"]
print("ln_tracks=$v", r["value"], "frames/s  LN ms/step", k["layernorm"]["ms_per_step"])
PY
done
done
for v in 0 1; do L4P_LN_TRACKS=$v python tools/prof_detail.py c3 5 2>/dev/null | grep "layernorm  *M131072"; done
