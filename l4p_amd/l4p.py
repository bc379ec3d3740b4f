"""L4PLitModule — host mirror of l4p/l4p.py (inference surface: forward / step("predict") / predict_step).

Subclasses lightning.LightningModule when lightning is importable (as the reference does), otherwise
torch.nn.Module; either way ``forward(batch, tasks)`` and ``predict_step(batch, batch_idx)`` behave as in
l4p.py:37-39,54-66,107-109.  Training hooks of the reference (losses/optimisers are ``None`` in the release)
are out of scope.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

try:  # pragma: no cover - lightning is not installed in the build image
    import lightning as L

    _Base = L.LightningModule
except Exception:  # noqa: BLE001
    _Base = torch.nn.Module


class L4PLitModule(_Base):
    def __init__(
        self,
        tasks: List[str],
        l4p_model: torch.nn.Module,
        loss_module: Optional[torch.nn.Module] = None,
        metrics_module: Optional[torch.nn.Module] = None,
        optimizer_opts: Optional[Dict[str, Any]] = None,
        scheduler_opts: Optional[Dict[str, Any]] = None,
        strict_loading: bool = True,
    ):
        super().__init__()
        self.tasks = tasks
        self.l4p_model = l4p_model
        self.loss_module = loss_module
        self.metrics_module = metrics_module
        self.optimizer_opts = optimizer_opts
        self.scheduler_opts = scheduler_opts
        self._strict = strict_loading

    def forward(self, batch, tasks):
        return self.l4p_model.forward(batch, tasks)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Lightning checkpoint layout: every key prefixed with ``l4p_model.`` (l4p.py:27)."""
        return self.l4p_model.load_state_dict(state_dict, strict=strict and self._strict)

    def step(self, phase, batch, batch_idx):
        dev = self.l4p_model.device
        for key in list(batch.keys()):
            if torch.is_tensor(batch[key]):
                batch[key] = batch[key].to(device=dev)
        out = self.forward(batch, self.tasks)
        if phase == "predict":
            return out
        raise NotImplementedError("only the inference (predict) phase is part of the MI355X engine")

    def predict_step(self, batch, batch_idx):
        return self.step("predict", batch, batch_idx)
