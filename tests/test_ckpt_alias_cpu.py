"""CPU: checkpoint ingest and the `l4p` import alias (row b).

* tools/ckpt_to_arena.py: a Lightning-format checkpoint written with torch.save -> packed arena file -> PackedWeights.load
  gives byte-identical tensors to packing the state_dict directly; schema violations are refused.
* `from l4p.models.utils import prepare_model` / `from l4p.data.video_dataset import VideoDataset` — the exact import lines
  of the reference's demo (/root/reference/demo/demo.py:13,16) — resolve to the engine's modules."""
import os
import subprocess
import sys

import pytest
import torch

from l4p_amd.packing import PackedWeights, pack_state_dict
from l4p_amd.weights import ModelCfg, seeded_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_demo_import_lines_resolve_to_the_engine():
    code = ("from l4p.models.utils import prepare_model\n"          # demo.py:13
            "from l4p.data.video_dataset import VideoDataset\n"      # demo.py:16
            "import l4p_amd.models.utils as u, l4p_amd.data.video_dataset as v\n"
            "assert prepare_model is u.prepare_model and VideoDataset is v.VideoDataset\n"
            "import inspect\n"
            "assert list(inspect.signature(prepare_model).parameters)[:5] == ['model_config_path', 'ckpt_path', 'max_queries', 'precision', 'accelerator']\n"
            "from l4p.l4p import L4PLitModule\n"
            "from l4p.models.l4p_videomae import L4P_VideoMAE\n"
            "from l4p.models.task_heads.dense_heads import VideoMAEDepthDPTHead\n"
            "from l4p.models.task_heads.sparse_heads import VideoMAETrack2DSamHead\n"
            "try:\n    import l4p.utils.vis\n    raise SystemExit('visualisation must not resolve')\nexcept ImportError:\n    pass\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env={**os.environ, "PYTHONPATH": ROOT})
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_ckpt_to_arena_roundtrip(tmp_path):
    from tools.ckpt_to_arena import convert

    cfg = ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    ckpt = tmp_path / "mini.ckpt"
    torch.save({"state_dict": {"l4p_model." + k: v for k, v in sd.items()}, "epoch": 3}, ckpt)  # Lightning layout
    for precision, td in (("bf16", torch.bfloat16), ("16-mixed", torch.float16), ("32-true", torch.float32)):
        out = tmp_path / f"mini.{precision}.l4parena"
        info = convert(str(ckpt), str(out), precision, cfg=cfg)
        assert PackedWeights.is_arena_file(str(out)) and not PackedWeights.is_arena_file(str(ckpt))
        got = PackedWeights.load(str(out), torch.device("cpu"))
        want = pack_state_dict(sd, cfg, td, torch.device("cpu"))
        assert [l[:1] + l[1:] for l in got.layout] == [l for l in want.layout]
        assert info["bytes"] == want.arena.numel() and got.extra["geometry"] == cfg.describe()
        for name, _, _, _ in want.layout:
            assert torch.equal(got[name].view(torch.uint8), want[name].view(torch.uint8)), name
    # the tool's default precision is prepare_model's / build_model's default (an arena packed with defaults loads with defaults),
    # every spelling the model accepts means the same dtype here, and an unknown string is an error (not float32)
    import inspect

    from l4p_amd.models import utils as mutils

    dflt = inspect.signature(convert).parameters["precision"].default
    assert dflt == inspect.signature(mutils.prepare_model).parameters["precision"].default == \
        inspect.signature(mutils.build_model).parameters["precision"].default
    info = convert(str(ckpt), str(tmp_path / "default.l4parena"), cfg=cfg)
    assert info["dtype"] == "float16"
    assert convert(str(ckpt), str(tmp_path / "x.l4parena"), "16", cfg=cfg)["dtype"] == "float16"
    with pytest.raises(ValueError):
        convert(str(ckpt), str(tmp_path / "x.l4parena"), "fp8", cfg=cfg)
    # a checkpoint with a missing / mis-shaped tensor is refused
    bad = dict(sd)
    bad.pop("video_encoder.norm.weight")
    torch.save({"state_dict": bad}, tmp_path / "bad.ckpt")
    with pytest.raises(SystemExit):
        convert(str(tmp_path / "bad.ckpt"), str(tmp_path / "bad.l4parena"), cfg=cfg)
