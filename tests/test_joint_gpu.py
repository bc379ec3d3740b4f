"""Rows a11 / f1 of SURVEY.md §8 on the GPU: the joint depth + camera seam alignment against its oracle
(oracle/joint_oracle.py, pinned against the imported reference by tools/gen_golden_joint.py).

Kernel level (csrc/umeyama.hip through the C ABI):
  l4p_quantile           == torch.quantile(x, q) — the reference's own call (aligner.py:187) — to float rounding
  l4p_point_map_samples  == generate_point_map (geometry_utils.py:13-53) at the engine's hashed pixel subset
  l4p_similarity_ransac  == the oracle's restatement of the engine's trial schedule (1e-3; inlier count within 0.5 %)
  l4p_similarity_apply   == KabaschUmeyama3DAligner.apply (aligner.py:239-265)
End to end: the engine's 3-window joint forward (mini geometry, depth + camray) == OracleModel(seam="engine"),
f32 engine 1e-3 relative-to-max, bf16 engine rel-L2."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd import _lib
from l4p_amd.weights import ModelCfg, seeded_state_dict
from oracle import joint_oracle as jo
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build, rel_l2


def _st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("n,q", [(8 * 224 * 224, 0.98), (1000, 0.98), (7, 0.5), (1, 0.98), (4096, 0.0), (4096, 1.0), (100003, 0.25)])
def test_quantile_equals_torch_quantile(dev, n, q):
    g = torch.Generator().manual_seed(n)
    x = torch.exp(torch.randn(n, generator=g))          # depth-like: positive, heavy tail
    if n > 100:
        x[torch.randint(0, n, (n // 3,), generator=g)] = float(x[5])  # many duplicates, also at the selected rank
    want = torch.quantile(x, q)
    lib = _lib.load()
    xd = x.cuda()
    ws = torch.empty(4100, dtype=torch.int32, device="cuda")
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    _lib.check(lib.l4p_quantile(_st(), xd.data_ptr(), n, q, ws.data_ptr(), out.data_ptr()), "l4p_quantile")
    got = float(out.cpu())
    assert abs(got - float(want)) <= 1e-6 * abs(float(want)), (got, float(want))


def test_quantile_of_duplicates_at_the_seam(dev):
    # rank lo and lo + 1 straddle two distinct values / fall on copies of one value
    lib = _lib.load()
    ws = torch.empty(4100, dtype=torch.int32, device="cuda")
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    for vals, q in (([1.0, 1.0, 2.0, 2.0, 3.0], 0.5), ([1.0, 2.0, 2.0, 2.0, 9.0], 0.6), ([0.0, 0.0, 5.0], 0.75), ([4.0, 1.0], 0.3)):
        x = torch.tensor(vals)
        _lib.check(lib.l4p_quantile(_st(), x.cuda().data_ptr(), len(vals), q, ws.data_ptr(), out.data_ptr()), "l4p_quantile")
        assert abs(float(out.cpu()) - float(torch.quantile(x, q))) <= 1e-6, (vals, q, float(out.cpu()))


def _scene(seed, F=3, H=224, W=224):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    depth = (2.0 + 0.5 * torch.sin(3 * xx + seed) + 0.3 * yy)[None, None, None].repeat(1, 1, F, 1, 1)
    depth = depth * (1.0 + 0.05 * torch.arange(F).float().view(1, 1, F, 1, 1)) + 0.01 * torch.rand(1, 1, F, H, W, generator=g)
    K = torch.eye(4)[None, :, :, None].repeat(1, 1, 1, F).clone()
    K[0, 0, 0], K[0, 1, 1], K[0, 0, 2], K[0, 1, 2] = 224.0, 220.0, 112.0, 111.0
    K[0, 0, 1] = 0.4
    P = torch.eye(4)[None, :, :, None].repeat(1, 1, 1, F).clone()
    for t in range(F):
        a = 0.05 * (t + 1)
        P[0, :3, :3, t] = torch.tensor([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1.0]])
        P[0, :3, 3, t] = torch.tensor([0.1 * t, -0.05 * t, 0.02 * t])
    return depth, K, P


def _engine_points(depth, K, P, ratio=10):
    lib = _lib.load()
    F, H, W = depth.shape[2], depth.shape[3], depth.shape[4]
    d = depth[0, 0].contiguous().cuda()
    Kf = K[0].permute(2, 0, 1).reshape(F, 16).contiguous().cuda()
    Pf = P[0].permute(2, 0, 1).reshape(F, 16).contiguous().cuda()
    o = torch.empty(F * ((H * W) // ratio), 3, dtype=torch.float32, device="cuda")
    _lib.check(lib.l4p_point_map_samples(_st(), d.data_ptr(), Kf.data_ptr(), Pf.data_ptr(), o.data_ptr(), F, H, W, ratio,
                                         jo.ENGINE_SEED), "l4p_point_map_samples")
    return o


def test_point_map_samples_equal_generate_point_map(dev):
    depth, K, P = _scene(1)
    F, H, W = 3, 224, 224
    got = _engine_points(depth, K, P).cpu()
    pm = jo.generate_point_map(depth, K, P)[0].reshape(3, F, H * W)            # reference arithmetic (pinned)
    sub = torch.from_numpy(jo.engine_pixel_subset(H, W, 10))
    want = pm[:, :, sub].permute(1, 2, 0).reshape(-1, 3)
    assert got.shape == want.shape
    assert (got - want).abs().max() <= 1e-5 * want.abs().max(), float((got - want).abs().max())
    # one sample per stride-10 cell, none repeated
    assert len(np.unique(sub.numpy())) == len(sub) == (H * W) // 10


def test_similarity_ransac_equals_oracle_schedule_and_apply(dev):
    depth_t, K, P_t = _scene(2)
    # the "current" window: the same scene through a similarity, plus 20 % gross depth outliers
    s_true = 1.3
    g = torch.Generator().manual_seed(9)
    depth_p = depth_t / s_true
    mask = torch.rand(depth_p.shape, generator=g) < 0.2
    depth_p = torch.where(mask, depth_p * (1.5 + torch.rand(depth_p.shape, generator=g)), depth_p)
    P_p = P_t.clone()
    P_p[0, :3, 3] = P_t[0, :3, 3] / s_true
    src, dst = _engine_points(depth_p, K, P_p), _engine_points(depth_t, K, P_t)
    n = src.shape[0]
    q98 = jo.depth_q98(depth_p)
    lib = _lib.load()
    trials = 100
    ws = torch.empty(15 * trials, dtype=torch.float32, device="cuda")
    out = torch.empty(18, dtype=torch.float32, device="cuda")
    _lib.check(lib.l4p_similarity_ransac(_st(), src.data_ptr(), dst.data_ptr(), n, q98.cuda().data_ptr(), jo.REPROJ_THRESHOLD, trials,
                                         jo.MIN_SAMPLES, jo.ENGINE_SEED, ws.data_ptr(), out.data_ptr()), "l4p_similarity_ransac")
    got = out.cpu().double().numpy()
    rel, inl = jo.engine_ransac(src.cpu().numpy(), dst.cpu().numpy(), float(q98[0]) * jo.REPROJ_THRESHOLD)
    assert np.abs(got[:16].reshape(4, 4) - rel["T"]).max() <= 1e-3 * np.abs(rel["T"]).max(), (got[:16].reshape(4, 4), rel["T"])
    assert abs(got[16] - rel["s"]) <= 1e-3 * rel["s"]
    assert abs(got[17] - inl.sum()) <= 0.005 * inl.sum() + 1, (got[17], int(inl.sum()))
    assert abs(got[16] - s_true) <= 2e-3 * s_true  # and both recover the ground truth
    # apply
    T = 16
    pose = torch.randn(1, 16, T, generator=g)
    dep = torch.rand(1, 1, T, 8, 8, generator=g) + 0.5
    sim = out.clone()
    pd, dd = pose[0].contiguous().cuda(), dep[0].contiguous().cuda()
    _lib.check(lib.l4p_similarity_apply(_st(), sim.data_ptr(), pd.data_ptr(), T, dd.data_ptr(), dd.numel()), "l4p_similarity_apply")
    relT = {"T": out[:16].view(1, 4, 4).cpu(), "s": out[16:17].cpu()}
    want = jo.similarity_apply(relT, {"camray": pose, "depth": dep})
    assert (pd.cpu() - want["camray"][0]).abs().max() <= 1e-5 * want["camray"].abs().max()
    assert (dd.cpu() - want["depth"][0]).abs().max() <= 1e-6 * want["depth"].abs().max()


_ORACLE_JOINT = {}


@pytest.mark.parametrize("precision", ["32-true", "bf16"])
def test_three_window_joint_forward_vs_oracle(dev, precision):
    """L4P_VideoMAE.forward -> joint_windowed_estimation (dense_heads.py:360-492) over 3 windows / 2 seams."""
    from oracle.l4p_oracle import OracleModel

    cfg = ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    model = build(cfg, sd, precision)
    batch = make_batch(32, 4)
    tasks = ["depth", "camray"]
    with torch.no_grad():
        out = model.forward({k: v.clone() for k, v in batch.items()}, tasks)
        if "om" not in _ORACLE_JOINT:  # (the CPU oracle's forward is the same for both engine precisions: once)
            om = OracleModel(sd, cfg, use_intrinsics=True, seam="engine")
            _ORACLE_JOINT["om"], _ORACLE_JOINT["ref"] = om, om.forward(batch, tasks)
        om, ref = _ORACLE_JOINT["om"], _ORACLE_JOINT["ref"]
    torch.cuda.synchronize()
    assert len(om.seam_log) == 2 and all(s["inliers"] >= 3 for s in om.seam_log), om.seam_log
    for key in ("depth_est_b1thw", "traj3d_est_b16t", "traj3d_intrinsics_est_b16t"):
        y, r = out[key].float().cpu(), ref[key]
        assert y.shape == r.shape, key
        if precision == "32-true":
            assert (y - r).abs().max() <= 1e-3 * r.abs().max(), (key, float((y - r).abs().max() / r.abs().max()))
    if precision == "32-true":
        return
    # bf16 engine: the per-window estimates drift by ~1e-2 from the f32 oracle's, and with random weights the two windows
    # of a seam are only loosely consistent, so the ESTIMATOR (best of 100 trials) legitimately lands elsewhere for
    # slightly different inputs.  The joint stage itself is therefore checked on the engine's OWN per-window estimates:
    # the oracle's seam flow fed with them must reproduce the engine's stitched result (the stage runs in f32 in both).
    ws = 16
    per_win = []
    with torch.no_grad():
        for st in (0, 8, 16):
            b = {k: (v[:, :, st:st + ws].clone() if k == "rgb_b3thw" else v[..., st:st + ws].clone() if k == "intrinsics_b44t" else v.clone())
                 for k, v in batch.items()}
            o = model.forward(b, tasks)
            per_win.append({"depth": o["depth_est_b1thw"].float().cpu(), "camray": o["traj3d_est_b16t"].float().cpu(),
                            "camray_intrinsics_est": o["traj3d_intrinsics_est_b16t"].float().cpu()})
    log = []
    est = jo.joint_windowed(lambda w: {k: v.clone() for k, v in per_win[w].items()}, [0, 8, 16], ws, "engine", log)
    for key, ek in (("depth_est_b1thw", "depth"), ("traj3d_est_b16t", "camray"), ("traj3d_intrinsics_est_b16t", "camray_intrinsics_est")):
        y, r = out[key].float().cpu(), est[ek]
        assert (y - r).abs().max() <= 1e-3 * r.abs().max(), (key, float((y - r).abs().max() / r.abs().max()), log)
    # the first window is never re-aligned: there the bf16 engine is within its drift of the f32 oracle
    assert rel_l2(out["depth_est_b1thw"][:, :, :8].float().cpu(), ref["depth_est_b1thw"][:, :, :8]) <= 3e-2
