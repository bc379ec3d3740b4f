"""Full geometry, bf16: the same forward twice (batch 1 and batch 4) - are the tracker outputs bit-identical run to run?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from l4p_amd.models.utils import build_model
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch

sd = seeded_state_dict(ModelCfg.full())
m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision="bf16")
m.load_state_dict({"l4p_model." + k: v for k, v in sd.items()})
KEYS = ("track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t")
bs = [make_batch(16, 8, seed=1234 + i) for i in range(4)]
b4 = {k: torch.cat([b[k] for b in bs], dim=0) for k in bs[0]}
for name, batch in (("batch 1", bs[2]), ("batch 4", b4)):
    outs = []
    for rep in range(3):
        with torch.no_grad():
            o = m.forward({k: v.clone() for k, v in batch.items()}, ["track_2d"])
        torch.cuda.synchronize()
        outs.append({k: o[k].float().cpu() for k in KEYS})
    for k in KEYS:
        print(name, k, "identical run to run:", all(torch.equal(outs[0][k], x[k]) for x in outs[1:]),
              "max diff", max(float((outs[0][k] - x[k]).abs().max()) for x in outs[1:]))
