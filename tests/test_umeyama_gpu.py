"""GPU: joint depth+pose seam alignment (row a11).  The reference step is skimage RANSAC on the CPU (randomised,
unpinned) => validated against SYNTHETIC GROUND TRUTH: a known similarity (s, R, t) between two windows'
overlap must be recovered (1e-3 relative) with 20 % gross outliers present, and applying it must map the
current window onto the stitched buffer."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd.models.aligner import KabaschUmeyama3DAligner


def _rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).float()


def test_similarity_recovered_with_outliers(dev):
    g = torch.Generator().manual_seed(3)
    ov, H, W = 8, 224, 224
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 224.0
    K[0, 2] = K[1, 2] = 112.0
    K44 = K[None, :, :, None].repeat(1, 1, 1, ov)
    # target ("already stitched") window: smooth depth + poses
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    depth_t = (2.0 + 0.5 * torch.sin(3 * xx) + 0.3 * yy)[None, None, None].repeat(1, 1, ov, 1, 1)
    depth_t = depth_t * (1.0 + 0.05 * torch.arange(ov).float().view(1, 1, ov, 1, 1))
    poses_t = torch.eye(4)[None, :, :, None].repeat(1, 1, 1, ov).clone()
    for t in range(ov):
        poses_t[0, :3, :3, t] = _rot(0.02 * t, -0.03 * t, 0.01 * t)
        poses_t[0, :3, 3, t] = torch.tensor([0.1 * t, -0.05 * t, 0.02 * t])
    # the current window sees the same scene through an unknown similarity: X_t = s R X_p + tr
    s_true, R_true, tr_true = 1.7, _rot(0.3, -0.2, 0.5), torch.tensor([0.4, -1.1, 0.8])
    # => depth_p = depth_t / s ; pose_p = S^-1 pose_t with rotation part re-normalised:  c2w_p = [R^T Rt | R^T (ct - tr)/s]
    depth_p = depth_t / s_true
    poses_p = torch.eye(4)[None, :, :, None].repeat(1, 1, 1, ov).clone()
    for t in range(ov):
        poses_p[0, :3, :3, t] = R_true.T @ poses_t[0, :3, :3, t]
        poses_p[0, :3, 3, t] = R_true.T @ (poses_t[0, :3, 3, t] - tr_true) / s_true
    # 20 % gross depth outliers in the current window
    mask = torch.rand(depth_p.shape, generator=g) < 0.2
    depth_p_noisy = torch.where(mask, depth_p * (1.5 + torch.rand(depth_p.shape, generator=g)), depth_p)
    pred = {"depth": depth_p_noisy.cuda(), "camray": poses_p.reshape(1, 16, ov).cuda(), "camray_intrinsics": K44.cuda()}
    target = {"depth": depth_t.cuda(), "camray": poses_t.reshape(1, 16, ov).cuda(), "camray_intrinsics": K44.cuda()}
    al = KabaschUmeyama3DAligner()
    al.solve(pred, target, (16, H, W))
    sim = al.rel_T_b44[0].cpu()
    T = sim[:16].view(4, 4)
    s_est = float(sim[16])
    assert abs(s_est - s_true) <= 1e-3 * s_true, s_est
    assert (T[:3, :3] / s_est - R_true).abs().max() <= 1e-3
    assert (T[:3, 3] - tr_true).abs().max() <= 2e-3
    assert float(sim[17]) > 0.7 * (3 * H * W // 10)  # ~80 % inliers
    cur = {"depth": depth_p.cuda(), "camray": poses_p.reshape(1, 16, ov).cuda(),
           "camray_intrinsics_est": K44.reshape(1, 16, ov).cuda()}
    new = al.apply(cur)
    assert (new["depth"].cpu() - depth_t).abs().max() <= 2e-3 * depth_t.abs().max()
    assert (new["camray"].cpu().view(1, 4, 4, ov) - poses_t).abs().max() <= 3e-3
    assert torch.equal(new["camray_intrinsics_est"].cpu(), cur["camray_intrinsics_est"].cpu())
