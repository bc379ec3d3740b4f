#!/bin/bash
# L4P_TRACK_BESIDE_STITCH=1 (the dense stitch beside the tail of the tracker recursion): the emulated-rank tests of the sharded long video
# (31 windows on 8 emulated ranks, dense outputs bit-identical to the single-GPU forward), the stream-overlap tests, and the full-length
# c5 bench line, several times over
export L4P_TRACK_BESIDE_STITCH=1
for i in 1 2 3 4; do
  python -m pytest tests/test_sharded_windows_gpu.py tests/test_stream_overlap_gpu.py tests/test_seam_local_gpu.py -x -q 2>&1 | tail -1
done
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from l4p_amd.parallel import forward_windows_sharded
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
tasks = list(bench.ALL_TASKS)
model, data, _ = bench.build_workload(tasks, 1, 64, dev, frames=256)
model.l4p_model.always_use_windowed_version = True
ref = None; bad = 0
for it in range(12):
    with torch.no_grad():
        out = forward_windows_sharded(model.l4p_model, data, tasks, 0, 1, group=4)
    torch.cuda.synchronize()
    cur = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
    if ref is None: ref = cur
    else:
        for k in ref:
            if not torch.equal(ref[k], cur[k]): bad += 1; print("MISMATCH", it, k, float((ref[k].float() - cur[k].float()).abs().max()))
print("full-size 256-frame video, 12 forwards with the stitch beside the tracker:", bad, "mismatching outputs")
PY
