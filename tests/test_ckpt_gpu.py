"""Row f2 on the GPU: prepare_model(model_config_path, ckpt_path, max_queries, precision) (l4p/models/utils.py:15-60) on a
REAL file in the reference's checkpoint format — torch.save({"state_dict": {l4p_model.*}}) — and on the packed arena
tools/ckpt_to_arena.py makes from it; both must give the same forward, bit for bit, as build_model + load_state_dict."""
import os
import shutil

import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd.models.utils import build_model, prepare_model
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, "configs", "model.yaml")
KEYS = ["depth_est_b1thw", "flow_2d_backward_est_b2thw", "track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "traj3d_est_b16t"]


def _forward(model, tasks, nq=5):
    batch = make_batch(16, nq)
    with torch.no_grad():
        out = model.forward({k: v.clone() for k, v in batch.items()}, tasks)
    torch.cuda.synchronize()
    return {k: out[k].clone() for k in KEYS if k in out}


@pytest.mark.parametrize("precision", ["bf16", "16-mixed", "32-true"])
def test_prepare_model_from_ckpt_and_arena_mini(dev, tmp_path, precision):
    from tools.ckpt_to_arena import convert

    cfg = ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    ckpt = str(tmp_path / "mini.ckpt")
    torch.save({"state_dict": {"l4p_model." + k: v for k, v in sd.items()}, "global_step": 1}, ckpt)
    tasks = ["depth", "flow_2d_backward", "track_2d", "camray"]

    def fix(m):  # hooks of the mini geometry (the yaml names the giant's 14/21/28/36)
        for h in m.l4p_model.task_heads.values():
            if hasattr(h, "hooks_idx"):
                h.hooks_idx = list(cfg.hooks)
        m.l4p_model.task_heads["camray"].use_intrinsics = True
        return m

    ref = fix(build_model(YAML, max_queries=3, precision=precision, model_cfg=cfg))
    ref.load_state_dict({"l4p_model." + k: v for k, v in sd.items()})
    want = _forward(ref, tasks)
    m1 = fix(prepare_model(YAML, ckpt, max_queries=3, precision=precision, model_cfg=cfg))
    assert m1.l4p_model.task_heads["track_2d"].max_queries == 3 and not m1.training
    got = _forward(m1, tasks)
    arena = str(tmp_path / "mini.l4parena")
    convert(ckpt, arena, precision, cfg=cfg)
    m2 = fix(prepare_model(YAML, arena, max_queries=3, precision=precision, model_cfg=cfg))
    got2 = _forward(m2, tasks)
    for k in want:
        assert torch.equal(got[k], want[k]), k
        assert torch.equal(got2[k], want[k]), k
    # an arena packed for the other dtype is refused
    other = "32-true" if precision == "bf16" else "bf16"
    with pytest.raises(ValueError):
        prepare_model(YAML, arena, precision=other, model_cfg=cfg)


def test_prepare_model_full_size_ckpt(dev, tmp_path):
    """The shipped geometry: a 916-key, 1.42 B-parameter fp32 checkpoint file (5.7 GB) through prepare_model exactly as
    demo.py:33-40 calls it; depth-only forward equals build_model + load_state_dict."""
    if shutil.disk_usage(str(tmp_path)).free < 9 * 2 ** 30:
        pytest.skip("needs 9 GB of scratch disk for the full-size checkpoint")
    cfg = ModelCfg.full()
    sd = seeded_state_dict(cfg)
    ckpt = str(tmp_path / "full.ckpt")
    torch.save({"state_dict": {"l4p_model." + k: v for k, v in sd.items()}}, ckpt)
    m = prepare_model(model_config_path=YAML, ckpt_path=ckpt, max_queries=128, precision="16-mixed", accelerator="gpu")
    got = _forward(m, ["depth"])
    del m
    torch.cuda.empty_cache()
    ref = build_model(YAML, max_queries=128, precision="16-mixed")
    ref.load_state_dict({"l4p_model." + k: v for k, v in sd.items()})
    want = _forward(ref, ["depth"])
    assert torch.equal(got["depth_est_b1thw"], want["depth_est_b1thw"])
    os.remove(ckpt)
