"""GPU parity at the FULL geometry (VideoMAE-v2-giant: 1408 wide, 40 blocks, all five heads) against the
golden vectors produced by the real reference (tests/golden/full_T16_all.npz).

L4P_F32 engine: 1e-3 relative-to-max on the sampled values (north_star).  L4P_BF16 engine: reported drift
(rel-L2 of samples) bounded loosely — 40 residual blocks in bf16 cannot meet 1e-3 (SURVEY.md §7 "hard parts").
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd.models.utils import build_model
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch, sample_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]


@pytest.fixture(scope="module")
def full_sd():
    return seeded_state_dict(ModelCfg.full())


def _run(sd, precision):
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=precision)
    m.l4p_model.task_heads["camray"].use_intrinsics = True
    m.load_state_dict({"l4p_model." + k: v for k, v in sd.items()})
    batch = make_batch(16, 8)
    with torch.no_grad():
        out = m.forward({k: v.clone() for k, v in batch.items()}, ALL)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("precision", ["32-true", "bf16"])
def test_full_size_all_heads_vs_reference_goldens(dev, full_sd, precision):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T16_all.npz"))
    out = _run(full_sd, precision)
    feats = out["enc_features_bpc_2dlist"][0]
    report = {}
    for li in (14, 21, 28, 36, 40):
        f = feats.f32(li).float().cpu().reshape(-1)
        s, g = f[sample_indices(f.numel())], torch.from_numpy(gold[f"feat{li}"])
        report[f"feat{li}"] = (float((s - g).abs().max() / g.abs().max()), float((s - g).norm() / g.norm()))
    for k in gold.files:
        if k.startswith("feat") or k.startswith("trace"):
            continue
        y = out[k].float().cpu().reshape(-1)
        g = torch.from_numpy(gold[k]).reshape(-1)
        s = y[sample_indices(y.numel())] if y.numel() > 4096 else y
        report[k] = (float((s - g).abs().max() / g.abs().max()), float((s - g).norm() / g.norm()))
    print(precision, {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in report.items()})
    if precision == "32-true":
        bad = {k: v for k, v in report.items() if v[0] > 1e-3}
        assert not bad, bad
    else:
        bad = {k: v for k, v in report.items() if v[1] > 0.15}
        assert not bad, bad
