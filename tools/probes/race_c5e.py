"""The seam-alignment kernels (fixed inputs, main stream) while the REAL tracker recursion runs on a side stream: mismatches?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "race_c5c.py")).read().split("ref = once()")[0])
from l4p_amd import parallel
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build

cfg = ModelCfg.mini()
model = build(cfg, seeded_state_dict(cfg), sys.argv[1] if len(sys.argv) > 1 else "bf16")
net = model.l4p_model
batch = make_batch(256, 2)
data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
with torch.no_grad():
    groups = parallel.encode_local_windows(net, data, ["track_2d"], 0, 1, 8)
    lasts = parallel.local_last_features(groups, 1)
    wins = [parallel.DecodedWindow(net.cfg.depth, {}, lasts[w]["last"]) for w in range(31)]
    trk = net.task_heads["track_2d"]
    strides = net.time_strides(256)
    ref = once()
    torch.cuda.synchronize()
    for mode in ("tracker on the same stream first", "tracker on a side stream (defer_join)"):
        bad = [0, 0, 0, 0, 0]
        for rep in range(6):
            trk.defer_join = mode.startswith("tracker on a side")
            trk.forward_windowed(enc_features_bpc_2dlist=wins, time_strides=strides, **data)
            for it in range(60):
                cur = once()
                for i, (r, c) in enumerate(zip(ref, cur)):
                    bad[i] += int(not torch.equal(r, c))
            trk.join_streams()
            trk.defer_join = False
            torch.cuda.synchronize()
        print(mode, "mismatches of (q98, pts a, pts b, sim, ransac ws) in 360 runs:", bad)
