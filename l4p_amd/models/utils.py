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
                  precision: str = "16-mixed", accelerator: str = "gpu"):
    """Same signature as the reference.  ``accelerator`` must be "gpu": there is no CPU path."""
    if accelerator not in ("gpu", "cuda", "auto"):
        raise ValueError("the MI355X engine only runs on the GPU (accelerator='gpu')")
    model = build_model(model_config_path, max_queries, precision)
    state_dict = torch.load(ckpt_path, weights_only=True)["state_dict"]
    model.load_state_dict(state_dict)
    return model.eval()
