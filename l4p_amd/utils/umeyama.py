"""GPU joint depth+pose seam alignment — host side of KabaschUmeyama3DAligner (aligner.py:158-265).

solve: q98 of the predicted overlap depth -> RANSAC threshold; point maps of every 3rd overlap frame for the
current window ("pred") and the already-stitched buffer ("target"), 10 % of the pixels; RANSAC similarity
(100 trials x 10 samples), re-estimated on the inliers.  apply: pose <- T pose (rotation / s), depth *= s.
Everything stays on the device; no host synchronisation.
"""
from __future__ import annotations

from typing import Dict

import torch

from .. import _lib
from ..ops import _p, _stream

SEED = 20250213


def _frames(x_b1thw_or_b16t: torch.Tensor, step: int) -> torch.Tensor:
    return x_b1thw_or_b16t[:, :, ::step]


def solve_window_similarity(cfg_obj, pred: Dict[str, torch.Tensor], target: Dict[str, torch.Tensor], img_info) -> torch.Tensor:
    """Returns float [B, 18]: row-major 4x4 similarity, scale, inlier count (device)."""
    lib = _lib.load()
    depth_p, depth_t = pred["depth"], target["depth"]
    B, _, ov, H, W = depth_p.shape
    dev = depth_p.device
    step = cfg_obj.frame_sample_step
    ratio = int(round(1.0 / cfg_obj.point_sample_ratio))
    out = torch.empty(B, 18, dtype=torch.float32, device=dev)
    ws_q = torch.empty(4100, dtype=torch.int32, device=dev)
    q98 = torch.empty(1, dtype=torch.float32, device=dev)
    trials = 100
    ws_r = torch.empty(15 * trials, dtype=torch.float32, device=dev)
    for b in range(B):
        dp = depth_p[b, 0].float().contiguous()
        _lib.check(lib.l4p_quantile(_stream(), _p(dp), dp.numel(), 0.98, _p(ws_q), _p(q98)), "l4p_quantile")
        pts = []
        for src, poses, Ks in ((depth_p, pred["camray"], pred["camray_intrinsics"]),
                               (depth_t, target["camray"], target["camray_intrinsics"])):
            d = src[b, 0, ::step].float().contiguous()                       # [F, H, W]
            F = d.shape[0]
            P = poses[b].reshape(4, 4, -1)[:, :, ::step].permute(2, 0, 1).reshape(F, 16).float().contiguous()
            K = Ks[b].reshape(4, 4, -1)[:, :, ::step].permute(2, 0, 1).reshape(F, 16).float().contiguous()
            o = torch.empty(F * ((H * W) // ratio), 3, dtype=torch.float32, device=dev)
            _lib.check(lib.l4p_point_map_samples(_stream(), _p(d), _p(K), _p(P), _p(o), F, H, W, ratio, SEED), "l4p_point_map_samples")
            pts.append(o)
        _lib.check(lib.l4p_similarity_ransac(_stream(), _p(pts[0]), _p(pts[1]), pts[0].shape[0], _p(q98),
                                             float(cfg_obj.reprojection_threshold), trials, int(cfg_obj.min_samples), SEED,
                                             _p(ws_r), out[b].data_ptr()), "l4p_similarity_ransac")
    return out


def apply_window_similarity(sim_b18: torch.Tensor, pred: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    lib = _lib.load()
    out = {}
    pose = pred["camray"].float().contiguous().clone()      # [B,16,T]
    depth = pred["depth"].float().contiguous().clone()      # [B,1,T,H,W]
    B, _, T = pose.shape
    for b in range(B):
        _lib.check(lib.l4p_similarity_apply(_stream(), sim_b18[b].data_ptr(), pose[b].data_ptr(), T, depth[b].data_ptr(),
                                            depth[b].numel()), "l4p_similarity_apply")
    out["camray"] = pose
    out["depth"] = depth
    for k, v in pred.items():
        if k not in out:
            out[k] = v
    return out
