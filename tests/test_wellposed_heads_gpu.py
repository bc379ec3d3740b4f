"""The head-level camera / joint-alignment path on a WELL-POSED problem at the real geometry (round-3 review, weak #2).

With random weights the decoded ray map is the image of no camera (consensus 8..13 of 256 rays) and neighbouring windows are
mutually inconsistent (1..7 inliers of 15 051 seam points), so the full-size goldens exercise the K estimate and the seam
RANSAC only in their degenerate regime.  Here the DECODER OUTPUTS are replaced at the head boundary (parallel.DecodedWindow:
what ``_decode`` would return) by renderings of a known camera path — ray maps ``[1,6,16,16,16]`` (get_rays_plucker semantics,
geometry_utils.py:165-241) and depth maps ``[1,1,16,224,224]`` of 4 overlapping windows, every window in its own frame (first
camera = identity) and its own scale, with measurement noise, and gross outliers in the depth — and the shipped configuration
(use_intrinsics=false, fixed_intrinsics=true, joint_alignment=true) runs from there: K from the first window's ray map
(csrc/intrinsics.hip), poses (geom.hip), 3 seams of RANSAC-Umeyama over 15 051 hashed points (umeyama.hip), chained apply.

Checked: (1) the K estimate has a real consensus (>= 200 of 256 rays) and equals the CPU restatement of the estimator
(oracle.l4p_oracle.engine_rays_to_intrinsics) and the TRUE K; (2) every seam keeps > 80 % of its points as inliers and the
engine's (T, s) equals the oracle's restatement of the schedule; (3) the stitched depth / poses / K over all 40 frames equal the
oracle's joint flow (joint_oracle.joint_windowed, pinned against the reference by tools/gen_golden_joint.py) on per-window
estimates computed by the oracle from the same ray maps, to 1e-3; (4) they equal the GROUND TRUTH (window 0's scale) to noise
level.  The geometry of this stage does not depend on the encoder width: the mini model carries it (shapes asserted)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd import parallel
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.test_intrinsics_gpu import render_rays
from tests.test_umeyama_gpu import _rot

H = W = 224
WS, STRIDE, T = 16, 8, 40
SCALES = [1.0, 1.7, 0.6, 1.25]


def _scene(seed=3):
    g = torch.Generator().manual_seed(seed)
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 231.0, 218.0, 108.0, 117.0
    c2w = torch.eye(4)[:, :, None].repeat(1, 1, T).clone()
    for t in range(1, T):
        c2w[:3, :3, t] = _rot(0.012 * t, -0.008 * t + 0.05 * np.sin(0.3 * t), 0.006 * t)
        c2w[:3, 3, t] = torch.tensor([0.04 * t, 0.015 * t * np.cos(0.2 * t), -0.025 * t])
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    depth = torch.stack([2.5 + 0.8 * torch.sin(4 * xx + 0.2 * t) * torch.cos(3 * yy - 0.1 * t) + 0.6 * yy + 0.02 * t
                         for t in range(T)])  # [T,H,W], 1.1 .. 4.7
    wins = []
    for w, st in enumerate(range(0, T - WS + 1, STRIDE)):
        s = SCALES[w]
        R0, o0 = c2w[:3, :3, st], c2w[:3, 3, st]
        rel = torch.eye(4)[:, :, None].repeat(1, 1, WS).clone()
        for j in range(WS):
            rel[:3, :3, j] = R0.T @ c2w[:3, :3, st + j]
            rel[:3, 3, j] = s * (R0.T @ (c2w[:3, 3, st + j] - o0))
        rays = render_rays(K, rel, H, W)
        rays = rays + 4e-4 * torch.randn(rays.shape, generator=g)
        d = s * depth[st:st + WS] * (1.0 + 1.5e-3 * torch.randn(WS, H, W, generator=g))
        bad = torch.rand(WS, H, W, generator=g) < 0.03  # gross outliers: the seam estimator has something to reject
        d = torch.where(bad, d * 1.4, d)
        wins.append({"rays": rays.float(), "depth": d[None, None].float(), "bad": bad, "rel": rel, "st": st})
    return K, c2w, depth, wins


def test_shipped_camera_and_joint_path_on_rendered_windows(dev):
    from oracle import joint_oracle as jo
    from oracle import l4p_oracle as lo
    from tests.test_encoder_dpt_gpu import build

    cfg = ModelCfg.mini()
    full = ModelCfg.full()
    assert (cfg.img, cfg.frames) == (full.img, full.frames)  # the stage below has the real geometry
    model = build(cfg, seeded_state_dict(cfg), "32-true")
    net = model.l4p_model
    cam = net.task_heads["camray"]
    cam.use_intrinsics = False  # as shipped (configs/model.yaml:44-45)
    assert cam.fixed_intrinsics is True and net.joint_alignment is True and tuple(cam.output_size) == (16, 16, 16)
    K, c2w, depth, wins = _scene()
    strides = net.time_strides(T)
    assert [int(s) for s in strides] == [w["st"] for w in wins]
    windows = [parallel.DecodedWindow(cfg.depth, {"depth": w["depth"].cuda(), "camray": w["rays"].cuda()}, None) for w in wins]
    data = {"rgb_b3thw": torch.zeros(1, 3, T, H, W, device="cuda"),
            "intrinsics_b44t": K[None, :, :, None].repeat(1, 1, 1, T).cuda()}
    seams = []
    from l4p_amd.models import aligner as al

    orig_solve = al.KabaschUmeyama3DAligner.solve

    def spy(self, pred, target, img_info):
        orig_solve(self, pred, target, img_info)
        seams.append(self.rel_T_b44.clone())

    al.KabaschUmeyama3DAligner.solve = spy
    try:
        with torch.no_grad():
            out = net.stitch_windows(windows, data, ["depth", "camray"], strides)
        torch.cuda.synchronize()
    finally:
        al.KabaschUmeyama3DAligner.solve = orig_solve
    dep = out["depth_est_b1thw"].float().cpu()
    pose = out["traj3d_est_b16t"].float().cpu()
    Kout = out["traj3d_intrinsics_est_b16t"].float().cpu()
    assert tuple(dep.shape) == (1, 1, T, H, W) and tuple(pose.shape) == (1, 16, T) and tuple(Kout.shape) == (1, 16, T)

    # ---- (1) the K estimate: a real consensus, == its CPU restatement, == the truth --------------------------------------------
    dirs = wins[0]["rays"][0, :3, 0].reshape(3, -1).T.numpy()
    want, n_cons, iters = lo.engine_rays_to_intrinsics(dirs, 16, 16, H, W, thr=0.2, b=0)
    got = Kout[0].reshape(4, 4, T)[..., 0].double().numpy()
    print(f"K estimate: consensus {n_cons}/256 after {iters} rounds; fx {got[0, 0]:.2f} fy {got[1, 1]:.2f} cx {got[0, 2]:.2f} cy {got[1, 2]:.2f}")
    assert n_cons >= 200, n_cons
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max(), (got, want)
    assert np.abs(got - K.double().numpy()).max() <= 5e-3 * 231.0, got
    assert (Kout[0] - Kout[0, :, :1]).abs().max() == 0  # one K for the whole clip

    # ---- per-window estimates by the ORACLE from the same ray maps (reference flow, K of the engine supplied for window 0) ----
    K_pix = Kout.reshape(1, 4, 4, T)[..., :WS]
    K_ray = lo.denormalize_intrinsics(lo.normalize_intrinsics(K_pix, H, W), 16, 16)[0, :3, :3, 0]
    K_in = data["intrinsics_b44t"].cpu()
    per_win = []
    for w, win in enumerate(wins):
        if w == 0:
            E, Kest = lo.rays_to_cameras_fixed_intrinsics(win["rays"], (H, W), k_override=lambda b: K_ray)
            first_K = Kest.clone()
        else:  # later windows rotate with the INPUT intrinsics and report the first window's estimate (dense_heads.py:327-333)
            E = lo.rays_to_cameras(win["rays"], lo.normalize_intrinsics(K_in[..., win["st"]:win["st"] + WS], H, W).float())
        p = torch.linalg.inv(E.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).reshape(1, 16, WS)
        per_win.append({"depth": win["depth"].clone(), "camray": p, "camray_intrinsics_est": first_K.reshape(1, 16, WS).clone()})

    # ---- (2) + (3): the joint flow of the oracle with the engine's deterministic draws ------------------------------------------
    log = []
    est = jo.joint_windowed(lambda w: {k: v.clone() for k, v in per_win[w].items()}, [int(s) for s in strides], WS, "engine", log)
    assert len(seams) == len(log) == 3
    for i, (s_eng, s_or) in enumerate(zip(seams, log)):
        row = s_eng[0].cpu().double().numpy()
        inl, n = int(round(row[17])), s_or["n"]
        print(f"seam {i}: engine inliers {inl}/{n}, oracle {s_or['inliers']}/{n}; scale {row[16]:.5f} vs {s_or['s']:.5f}")
        assert n == 15051 and inl >= 0.8 * n and s_or["inliers"] >= 0.8 * n, (i, inl, s_or["inliers"])
        assert abs(inl - s_or["inliers"]) <= 0.005 * n
        assert abs(row[16] - s_or["s"]) <= 1e-3 * s_or["s"]
        assert np.abs(row[:16].reshape(4, 4) - s_or["T"]).max() <= 1e-3 * np.abs(s_or["T"]).max()
    for key, ek in (("depth_est_b1thw", "depth"), ("traj3d_est_b16t", "camray"), ("traj3d_intrinsics_est_b16t", "camray_intrinsics_est")):
        y, r = out[key].float().cpu(), est[ek]
        e = float((y - r).abs().max() / r.abs().max())
        print(f"{key}: engine vs oracle joint flow {e:.2e}")
        assert e <= 1e-3, (key, e)

    # ---- (4) ground truth: everything lands in window 0's frame and scale ------------------------------------------------------
    s0 = SCALES[0]
    last_writer = [max(w for w, win in enumerate(wins) if win["st"] <= t < win["st"] + WS) for t in range(T)]
    good = torch.stack([~wins[last_writer[t]]["bad"][t - wins[last_writer[t]]["st"]] for t in range(T)])
    rel_d = (dep[0, 0] / (s0 * depth) - 1.0).abs()
    print(f"depth vs truth: median {float(rel_d[good].median()):.2e}, p99 {float(rel_d[good].quantile(0.99)):.2e}")
    assert float(rel_d[good].quantile(0.99)) <= 1.5e-2
    P = pose[0].reshape(4, 4, T)
    Rerr = max(float((P[:3, :3, t] - c2w[:3, :3, t]).abs().max()) for t in range(T))
    terr = max(float((P[:3, 3, t] - s0 * c2w[:3, 3, t]).abs().max()) for t in range(T))
    print(f"poses vs truth: rotation {Rerr:.2e}, translation {terr:.2e} (path length {float(c2w[:3, 3, -1].norm()):.2f})")
    assert Rerr <= 1e-2 and terr <= 2e-2 * float(c2w[:3, 3, -1].norm())
