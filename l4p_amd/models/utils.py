"""prepare_model — host mirror of l4p/models/utils.py:15-60.

The reference instantiates configs/model.yaml through jsonargparse and wraps the module with Lightning
Fabric.  Neither is required here: the ``class_path`` / ``init_args`` tree is instantiated by a small
recursive loader that maps the reference's class paths (``l4p.…``) onto this package (``l4p_amd.…``),
and "Fabric.setup" reduces to choosing the engine dtype and device.
"""
from __future__ import annotations

import importlib
from typing import Any, Optional

import torch
import yaml


def _resolve(class_path: str):
    if class_path.startswith("l4p."):
        class_path = "l4p_amd." + class_path[len("l4p."):]
    mod, _, name = class_path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node: Any, **overrides) -> Any:
    """jsonargparse-style ``{class_path, init_args}`` tree -> objects (lists/dicts are walked recursively)."""
    if isinstance(node, dict) and "class_path" in node:
        kwargs = {k: instantiate(v) for k, v in (node.get("init_args") or {}).items()}
        kwargs.update(overrides)
        return _resolve(node["class_path"])(**kwargs)
    if isinstance(node, dict):
        return {k: instantiate(v) for k, v in node.items()}
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    return node


def build_model(model_config_path: str, max_queries: Optional[int] = None, precision: str = "16-mixed", model_cfg=None):
    with open(model_config_path, "r") as f:
        model_dict = yaml.safe_load(f)
    l4p_node = model_dict["init_args"]["l4p_model"]
    if max_queries is not None:
        l4p_node["init_args"]["task_heads"]["init_args"]["modules"]["track_2d"]["init_args"]["max_queries"] = max_queries
    l4p_node["init_args"]["precision"] = precision
    if model_cfg is not None:
        l4p_node["init_args"]["model_cfg"] = model_cfg
    return instantiate(model_dict)


def prepare_model(model_config_path: str, ckpt_path: Optional[str], max_queries: Optional[int] = None,
                  precision: str = "16-mixed", accelerator: str = "gpu", model_cfg=None):
    """Same signature as the reference (l4p/models/utils.py:15-60).  ``accelerator`` must be "gpu": there is no CPU path.
    ``ckpt_path``: the reference's Lightning checkpoint ({"state_dict": {916 keys prefixed l4p_model.}}, :52-53), or a
    packed arena written offline by tools/ckpt_to_arena.py (recognised by its magic; skips the repacking at start-up).
    ``model_cfg`` (engine extension, keyword only in practice): a non-default geometry, used by the tests' mini model."""
    if accelerator not in ("gpu", "cuda", "auto"):
        raise ValueError("the MI355X engine only runs on the GPU (accelerator='gpu')")
    from ..packing import PackedWeights

    if ckpt_path is None:
        # (the reference's Optional[str] annotation notwithstanding, it has no weights-free mode either: torch.load(None) fails)
        raise ValueError("prepare_model needs ckpt_path: the reference's Lightning checkpoint or a packed arena "
                         "(tools/ckpt_to_arena.py); build_model(...) gives a model without weights")
    model = build_model(model_config_path, max_queries, precision, model_cfg=model_cfg)
    if PackedWeights.is_arena_file(ckpt_path):
        net = model.l4p_model
        if not torch.cuda.is_available():
            from .. import _lib

            raise _lib.L4PHipError("no AMD GPU visible: the L4P engine has no CPU path")
        pw = PackedWeights.load(ckpt_path, torch.device("cuda", torch.cuda.current_device()))
        want = {0: "bfloat16", 1: "float32", 2: "float16"}[net.engine_dtype]
        have = getattr(pw, "extra", {}).get("dtype")
        if have != want:
            flag = {"bfloat16": "bf16", "float16": "16-mixed", "float32": "32-true"}
            raise ValueError(f"{ckpt_path} was packed for {have}, the model was built with precision={precision!r} ({want}): pass "
                             f"precision={flag.get(have, have)!r} here (demo/demo.py --precision {flag.get(have, have)}), or repack with "
                             f"tools/ckpt_to_arena.py --precision {flag[want]}")
        if getattr(pw, "extra", {}).get("geometry") != net.cfg.describe():
            raise ValueError(f"{ckpt_path} was packed for another model geometry")
        net.set_weights(pw)
        return model.eval()
    state_dict = torch.load(ckpt_path, weights_only=True)["state_dict"]
    model.load_state_dict(state_dict)
    return model.eval()
