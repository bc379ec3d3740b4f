#!/bin/bash
# Scaling curve on ONE node with N GPUs (BASELINE.json configs[2]/[3]/[4]):
#   tools/scale.sh [max_gpus=8] [steps=10] [warmup=3]
# c3 at 1 GPU (configs[2], batch 4) and at 2/4/8 GPUs (configs[3]: 8 clips per GPU, weak scaling, RCCL weight broadcast once,
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
for wl in ("configs[2]", "configs[3]", "configs[4]"):
    for r in rows:
        if r["config"]["workload"].startswith(wl):
            print(f'{wl} n={r["n_gpus"]} {r["value"]:.1f} frames/s  {r["ms_per_step"]:.1f} ms/step  rccl_ranks={r.get("rccl_ranks")}')
PY
