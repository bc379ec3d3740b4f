#!/bin/bash
# Scaling curve on ONE node with N GPUs (BASELINE.json configs[2]/[3]/[4]):
#   tools/scale.sh [max_gpus=8] [steps=10] [warmup=3]
# c3 at 1 GPU (configs[2], batch 4; and once at batch 8 = the per-GPU batch of the N >= 2 runs, so that the curve is
# like-for-like) and at 2/4/8 GPUs (configs[3]: 8 clips per GPU, weak scaling, RCCL weight broadcast once,
# no collective in the step), then c5 (configs[4]: one 256-frame video, windows sharded, strong scaling).  One JSON line per
# run is appended to gpurun_out/scale.jsonl; every N > 1 line carries "rccl_ranks" = N from the live collective self-test.
set -u
cd "$(dirname "$0")/.."
MAXG=${1:-8}; STEPS=${2:-10}; WARM=${3:-3}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
run() {  # n, extra bench args...
  local n=$1; shift
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM "$@"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps $STEPS --warmup $WARM "$@"
  fi
}
# like-for-like base of the weak-scaling curve: one GPU at configs[3]'s per-GPU batch (8 clips), the batch the N >= 2 runs use
echo "== c3, 1 GPU, batch 8 (the per-GPU batch of configs[3]: base of the efficiency figures)" >&2
run 1 --batch 8 --no-cpu-baseline | tail -1 | tee -a gpurun_out/scale.jsonl
for n in 1 2 4 8; do
  [ "$n" -gt "$MAXG" ] && break
  echo "== c3, $n GPU(s)" >&2
  run $n --no-cpu-baseline | tail -1 | tee -a gpurun_out/scale.jsonl
done
for n in 1 2 4 8; do
  [ "$n" -gt "$MAXG" ] && break
  echo "== c5 (256 frames), $n GPU(s)" >&2
  run $n --workload c5 --no-cpu-baseline | tail -1 | tee -a gpurun_out/scale.jsonl
done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/scale.jsonl") if l.strip().startswith("{")]
base = {}
for r in rows:
    wl = r["config"]["workload"][:10]
    key = (wl, r["n_gpus"], r["config"].get("clips_per_gpu_per_step"))
    if r["n_gpus"] == 1:
        base[(wl, r["config"].get("clips_per_gpu_per_step"))] = r["value"]
for wl in ("configs[2]", "configs[3]", "configs[4]"):
    for r in rows:
        if r["config"]["workload"].startswith(wl):
            b = r["config"].get("clips_per_gpu_per_step")
            # efficiency against the one-GPU line with the SAME per-GPU batch (configs[3] against "configs[2]" at batch 8)
            ref = base.get((wl, b)) or base.get(("configs[2]", b))
            eff = f'  efficiency {r["value"] / (ref * r["n_gpus"]):.3f} vs 1 GPU at batch {b}' if ref else ""
            print(f'{wl} n={r["n_gpus"]} batch/GPU={b} {r["value"]:.1f} frames/s  {r["ms_per_step"]:.1f} ms/step  rccl_ranks={r.get("rccl_ranks")}{eff}')
PY
