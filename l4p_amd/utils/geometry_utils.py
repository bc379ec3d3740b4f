"""Pose geometry on the GPU — host mirror of the parts of l4p/utils/geometry_utils.py on the hot path."""
from __future__ import annotations

import torch

from .. import _lib
from ..ops import _p, _stream


def normalize_intrinsics(intrinsics_b44t: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """geometry_utils.py:110-116 (kept for callers; the pose kernel re-scales intrinsics itself)."""
    K = intrinsics_b44t.clone().detach()
    K[:, :2, 2] += 0.5
    K[:, 0] = K[:, 0] / w
    K[:, 1] = K[:, 1] / h
    return K


def poses_from_rays(rays_b6thw: torch.Tensor, intrinsics_b44t: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """rays_to_cameras(rays, normalize_intrinsics(K, H, W)) followed by torch.linalg.inv and the b16t reshape
    (geometry_utils.py:331-406; dense_heads.py:322-348) as one kernel: float [B,6,T,h,w] -> float [B,16,T]."""
    assert rays_b6thw.is_cuda and rays_b6thw.dtype == torch.float32
    B, six, T, h, w = rays_b6thw.shape
    assert six == 6 and tuple(intrinsics_b44t.shape) == (B, 4, 4, T)
    rays = rays_b6thw.contiguous()
    K = intrinsics_b44t.to(device=rays.device, dtype=torch.float32).contiguous()
    out = torch.empty(B, 16, T, dtype=torch.float32, device=rays.device)
    _lib.check(_lib.load().l4p_rays_to_pose(_stream(), _p(rays), _p(K), _p(out), B, T, h, w, H, W), "l4p_rays_to_pose")
    return out


def cameras_from_rays_variable_intrinsics(rays_b6thw: torch.Tensor, H: int, W: int, reproj_threshold: float = 0.2):
    """rays_to_cameras_and_variable_per_frame_intrinsics (geometry_utils.py:582-654) + the pose inversion of
    dense_heads.py:346-348: every frame's own K and the rotation of H^-1 = K R as that frame's camera rotation.
    float [B,6,T,h,w] -> (world_T_cam float [B,16,T], K float [B,4,4,T] in pixel units of the H x W image)."""
    assert rays_b6thw.is_cuda and rays_b6thw.dtype == torch.float32
    B, six, T, h, w = rays_b6thw.shape
    rays = rays_b6thw.contiguous()
    K = torch.empty(B, 4, 4, T, dtype=torch.float32, device=rays.device)
    R = torch.empty(B, 9, T, dtype=torch.float32, device=rays.device)
    pose = torch.empty(B, 16, T, dtype=torch.float32, device=rays.device)
    lib = _lib.load()
    _lib.check(lib.l4p_rays_to_intrinsics_frames(_stream(), _p(rays), _p(K), _p(R), None, B, T, h, w, H, W, reproj_threshold),
               "l4p_rays_to_intrinsics_frames")
    _lib.check(lib.l4p_rays_to_pose_rot(_stream(), _p(rays), _p(R), _p(pose), B, T, h, w), "l4p_rays_to_pose_rot")
    return pose, K


def intrinsics_from_rays(rays_b6thw: torch.Tensor, H: int, W: int, reproj_threshold: float = 0.2, frame: int = 0) -> torch.Tensor:
    """Fixed intrinsics from the first frame's ray map (rays_to_cameras_and_fixed_per_frame_intrinsics,
    geometry_utils.py:493-579, K part): float [B,6,T,h,w] -> float [B,4,4,T] in pixel units of the H x W image."""
    assert rays_b6thw.is_cuda and rays_b6thw.dtype == torch.float32
    B, six, T, h, w = rays_b6thw.shape
    rays = rays_b6thw.contiguous()
    K = torch.empty(B, 4, 4, T, dtype=torch.float32, device=rays.device)
    _lib.check(_lib.load().l4p_rays_to_intrinsics(_stream(), _p(rays), _p(K), None, B, T, h, w, H, W, frame, reproj_threshold),
               "l4p_rays_to_intrinsics")
    return K
