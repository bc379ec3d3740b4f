"""SAM-style point tracker on the MI355X engine — host mirror of l4p/models/task_heads/sparse_heads.py."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch


class VideoMAETrack2DSamHead(torch.nn.Module):
    def __init__(self, task_name: str = "track_2d", prompt_embed_dim: int = 1408,
                 image_size: Tuple[int, int, int] = (16, 224, 224), patch_size: Tuple[int, int, int] = (2, 14, 14),
                 estimate_vis: bool = False, estimate_depth: bool = False, sam_head_depth: int = 2,
                 decoding_out_dim_factor: int = 8, num_prompt_points: int = 2, num_point_embeddings: int = 2,
                 modify_pointlabels_for_windowing: bool = False, prompt_using_features: bool = False,
                 attend_to_past: bool = False, depth_fn: str = "linear", vis_fn: str = "linear",
                 estimation_directions: List[int] = [1, -1], max_queries: int = 192):
        super().__init__()
        self.task_name = task_name
        self.max_queries = max_queries
        self._rt = None
        self._engine_task = ""

    def forward_windowed(self, *a, **k):
        raise NotImplementedError("tracker head: under construction")
