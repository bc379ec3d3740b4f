"""Per-track distance of the bf16 tracker from the f32 tracker, folded vs projected image -> token attention (Python composition)."""
import os
import sys

os.environ["L4P_TRACK_PYTHON"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build

cfg = ModelCfg.mini()
sd = seeded_state_dict(cfg)
nq, T = int(sys.argv[1]) if len(sys.argv) > 1 else 9, int(sys.argv[2]) if len(sys.argv) > 2 else 40
batch = make_batch(T, nq)
out = {}
for prec in ("32-true", "bf16"):
    model = build(cfg, sd, prec)
    for fold in ("1", "0"):
        os.environ["L4P_TRACK_FOLD_I2T"] = fold
        with torch.no_grad():
            o = model.forward({k: v.clone() for k, v in batch.items()}, ["track_2d"])
        out[(prec, fold)] = {k: v.float().cpu() for k, v in o.items() if torch.is_tensor(v)}
    del model
ref = out[("32-true", "0")]
for k in ("track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t"):
    for fold in ("1", "0"):
        d = (out[("bf16", fold)][k] - ref[k])[0]
        per = d.flatten(1).norm(dim=1) / ref[k][0].flatten(1).norm(dim=1).clamp_min(1e-9)
        print(k, "folded" if fold == "1" else "projected", "per-track rel-L2 vs f32:", " ".join(f"{x:.1e}" for x in per.tolist()))
