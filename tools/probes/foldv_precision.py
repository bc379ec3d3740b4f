"""Full geometry: per-track distance of the bf16 tracker from the f32 tracker with the value projection of the token -> image
attention folded (L4P_TRACK_FOLD_T2I_V=1) vs projected (=0), Python composition (the switch is read per call), and the distance
between a batch-1 and a batch-2 evaluation of the same clip in each form (what tests/test_full_model_gpu.py gates).
usage: foldv_precision.py [queries=16] [frames=16]"""
import os
import sys

os.environ["L4P_TRACK_PYTHON"] = "1"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from l4p_amd.models.utils import build_model
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch

sd = seeded_state_dict(ModelCfg.full())
nq, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 16
batch = make_batch(T, nq)
other = make_batch(T, nq, seed=77)
both = {k: torch.cat([batch[k], other[k]], dim=0) for k in batch}
KEYS = ("track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t")
out = {}
for prec in ("32-true", "bf16"):
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=prec)
    m.load_state_dict({"l4p_model." + k: v for k, v in sd.items()})
    for fold in (["0"] if prec == "32-true" else ["1", "0"]):
        os.environ["L4P_TRACK_FOLD_T2I_V"] = fold
        with torch.no_grad():
            o1 = m.forward({k: v.clone() for k, v in batch.items()}, ["track_2d"])
            o2 = m.forward({k: v.clone() for k, v in both.items()}, ["track_2d"]) if prec == "bf16" else None
        out[(prec, fold)] = ({k: o1[k].float().cpu() for k in KEYS}, None if o2 is None else {k: o2[k].float().cpu() for k in KEYS})
    del m
ref = out[("32-true", "0")][0]


def per_track(x, y):
    return (x - y).flatten(1).norm(dim=1) / y.flatten(1).norm(dim=1).clamp_min(1e-9)


for k in KEYS:
    for (prec, fold), (o1, o2) in out.items():
        if prec != "bf16":
            continue
        d = per_track(o1[k][0], ref[k][0])
        b = per_track(o2[k][0], o1[k][0])
        print(f"{k:28s} FOLD_V={fold:3s} vs f32: median {float(d.median()):.1e} max {float(d.max()):.1e} all {float((o1[k][0] - ref[k][0]).norm() / ref[k][0].norm()):.1e}"
              f" | batch 2 vs batch 1: median {float(b.median()):.1e} max {float(b.max()):.1e} all {float((o2[k][0] - o1[k][0]).norm() / o1[k][0].norm()):.1e}")
