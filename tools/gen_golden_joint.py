"""Pin oracle/joint_oracle.py against the REAL reference (runs only where /root/reference exists) and write
tests/golden/mini_T32_joint.npz.

  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_joint.py

1. Function level, imported reference vs oracle on seeded inputs (asserted here, recorded in the fixture's report):
   generate_point_map (geometry_utils.py:13-53), the q98 threshold (aligner.py:187-188), KabaschUmeyama3DAligner.apply
   (aligner.py:239-265).
2. Flow level: the reference's own L4P_VideoMAE.forward -> joint_windowed_estimation (dense_heads.py:360-492) on the
   mini geometry, 32 frames = 3 windows = 2 seams, tasks depth + camray, with its two RANDOM draws replaced by fixed
   stand-ins injected from oracle/joint_oracle.py: np.random.permutation -> fixed_permutation (aligner.py:216-220) and
   get_similarity_3d_transform -> fixed_inlier_estimator (closed-form Umeyama on a fixed inlier set instead of
   skimage.measure.ransac, aligner.py:139-146).  Everything else — point maps, quantile, sub-sampling, apply, stitching,
   pose chaining — is reference code.  The oracle's joint_windowed(seam="fixed") must reproduce it (asserted <= 1e-4).
Only data is written: sampled outputs, the per-seam thresholds / transforms the reference computed, and the report.
"""
from __future__ import annotations

import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from l4p_amd.weights import ModelCfg, seeded_state_dict
from oracle import joint_oracle as jo
from tests.golden_utils import make_batch, sample_indices
from tools.gen_golden import build_reference, install_stubs, rel_err


def main():
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    import l4p.models.aligner as ref_al
    from l4p.utils.geometry_utils import generate_point_map as ref_pm

    report = {}
    g = torch.Generator().manual_seed(77)
    # ---- 1. function level -------------------------------------------------------------------------------------
    B, T, H, W = 2, 3, 48, 64
    depth = torch.rand(B, 1, T, H, W, generator=g) * 3 + 0.2
    K = torch.eye(4).view(1, 4, 4, 1).repeat(B, 1, 1, T).clone()
    K[:, 0, 0] = 60.0 + torch.rand(B, T, generator=g)
    K[:, 1, 1] = 55.0 + torch.rand(B, T, generator=g)
    K[:, 0, 1] = 0.3  # skew: the general 3x3 inverse is exercised
    K[:, 0, 2], K[:, 1, 2] = 31.5, 23.5
    A = torch.randn(B, T, 3, 3, generator=g)
    R = torch.linalg.qr(A).Q
    P = torch.eye(4).view(1, 1, 4, 4).repeat(B, T, 1, 1).clone()
    P[:, :, :3, :3] = R
    P[:, :, :3, 3] = torch.randn(B, T, 3, generator=g)
    P = P.permute(0, 2, 3, 1).contiguous()
    e = rel_err(jo.generate_point_map(depth, K, P), ref_pm(depth, K, P))
    report["generate_point_map_rel_err"] = e
    assert e <= 1e-6, e
    # q98: the reference's expression, verbatim semantics (aligner.py:187)
    q_ref = torch.quantile(depth.reshape(B, -1).to(torch.float32), 0.98, dim=-1)
    assert torch.equal(jo.depth_q98(depth), q_ref)
    # apply
    al = ref_al.KabaschUmeyama3DAligner()
    s = torch.tensor([1.7, 0.6])
    Tm = torch.eye(4).repeat(B, 1, 1)
    Tm[:, :3, :3] = s.view(B, 1, 1) * torch.linalg.qr(torch.randn(B, 3, 3, generator=g)).Q
    Tm[:, :3, 3] = torch.randn(B, 3, generator=g)
    al.rel_T_b44 = {"T": Tm, "s": s[0:1]}  # the reference broadcasts s over the batch as a [bs] tensor; bs = 1 in its flow
    cur = {"depth": depth[:1], "camray": P[:1].reshape(1, 16, T).clone(), "camray_intrinsics_est": K[:1].reshape(1, 16, T)}
    al.rel_T_b44 = {"T": Tm[:1], "s": s[:1]}
    want = al.apply({k: v.clone() for k, v in cur.items()})
    got = jo.similarity_apply({"T": Tm[:1], "s": s[:1]}, cur)
    for k in want:
        e = rel_err(got[k], want[k])
        report[f"apply_{k}_rel_err"] = e
        assert e <= 1e-6, (k, e)

    # LinearAligner(pre_post_fn="inverse", method="mean") (aligner.py:69-118) and LstSqAffineAligner("inverse") (:29-66)
    from oracle import l4p_oracle as lo

    pr = torch.rand(2, 1, 8, 24, 24, generator=g) * 3 + 0.3
    tg = pr * 1.3 + 0.05 * torch.randn(pr.shape, generator=g)
    pr[0, 0, 0, 0, :4] = 0.0   # invalid depths: safe_inverse -> 0 on both sides
    tg[1, 0, 1, 2, :3] = -1.0
    la = ref_al.LinearAligner(pre_post_fn="inverse", method="mean")
    la.solve(pr, tg, None, None)
    e = rel_err(lo.linear_mean_solve(pr, tg), la.sol)
    report["linear_aligner_solve_rel_err"] = e
    assert e <= 1e-6, e
    e = rel_err(lo.linear_mean_apply(pr, lo.linear_mean_solve(pr, tg)), la.apply(pr))
    report["linear_aligner_apply_rel_err"] = e
    assert e <= 1e-6, e
    for pp in ("inverse", "identity"):
        lm = ref_al.LinearAligner(pre_post_fn=pp, method="median")
        lm.solve(pr, tg - (0.0 if pp == "inverse" else 2.0), None, None)  # identity: ratios of either sign
        got = lo.linear_median_solve(pr, tg - (0.0 if pp == "inverse" else 2.0), inverse=pp == "inverse")
        assert torch.equal(got, lm.sol), (pp, got, lm.sol)
        report[f"linear_aligner_median_{pp}_abs_diff"] = float((got - lm.sol).abs().max())
    aa = ref_al.LstSqAffineAligner(pre_post_fn="inverse")
    aa.solve(pr, tg, None, None)
    e = rel_err(lo.lstsq_affine_apply(pr, lo.lstsq_affine_solve(pr, tg)), aa.apply(pr))
    report["affine_aligner_apply_rel_err"] = e
    assert e <= 1e-5, e

    # ---- 2. the 3-window flow through the reference ---------------------------------------------------------------
    cfg = ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    model = build_reference(cfg)
    model.load_state_dict(sd, strict=True)
    batch = make_batch(32, 4)
    seams = []

    def spy_estimator(src, dst, min_samples=5, reprojection_threshold=0.1, confidence=0.99):
        rel, inl = jo.fixed_inlier_estimator(src, dst, min_samples, reprojection_threshold, confidence)
        seams.append({"thr": float(reprojection_threshold), "T": rel["T"].copy(), "s": float(rel["s"]), "n": int(src.shape[0])})
        return rel, inl

    orig_est, orig_perm = ref_al.get_similarity_3d_transform, np.random.permutation
    ref_al.get_similarity_3d_transform = spy_estimator
    np.random.permutation = jo.fixed_permutation
    try:
        with torch.no_grad():
            out = model.forward({k: v.clone() for k, v in batch.items()}, ["depth", "camray"])
    finally:
        ref_al.get_similarity_3d_transform, np.random.permutation = orig_est, orig_perm
    assert len(seams) == 2 and seams[0]["n"] == int(0.1 * 3 * 224 * 224)

    from oracle.l4p_oracle import OracleModel

    om = OracleModel(sd, cfg, use_intrinsics=True, seam="fixed")
    with torch.no_grad():
        oout = om.forward(batch, ["depth", "camray"])
    npz = {}
    for k in ("depth_est_b1thw", "traj3d_est_b16t", "traj3d_intrinsics_est_b16t"):
        e = rel_err(oout[k], out[k])
        report[f"flow_{k}_rel_err"] = e
        assert e <= 1e-4, (k, e)
        v = out[k].detach().float()
        npz[k] = v.reshape(-1)[sample_indices(v.numel())].numpy() if v.numel() > 4096 else v.numpy()
    for i, (a, b) in enumerate(zip(seams, om.seam_log)):
        assert abs(a["thr"] - b["thr"]) <= 1e-6 * abs(a["thr"]), (a["thr"], b["thr"])
        assert np.abs(a["T"] - b["T"]).max() <= 1e-5 * np.abs(a["T"]).max()
        npz[f"seam{i}_thr"] = np.float32(a["thr"])
        npz[f"seam{i}_T"] = a["T"].astype(np.float64)
        npz[f"seam{i}_s"] = np.float64(a["s"])
    out_dir = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(out_dir, "mini_T32_joint.npz"), **npz)
    with open(os.path.join(out_dir, "oracle_vs_reference_joint.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
