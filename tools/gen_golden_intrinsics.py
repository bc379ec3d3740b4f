"""Pin the oracle's restatement of the K-estimation path — the shipped default `use_intrinsics: false, fixed_intrinsics: true`
(configs/model.yaml:44-45) — against the REAL reference (runs only where /root/reference exists) and write
tests/golden/mini_T32_default_config.npz.

  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_intrinsics.py

The reference estimates K with cv2.findHomography(RANSAC) + cv2.RQDecomp3x3 (geometry_utils.py:436-448; opencv unpinned and not
installed here).  Those two calls are replaced ON BOTH SIDES by the deterministic stand-ins of oracle/l4p_oracle.py
(dlt_homography, rq3), installed into the reference's cv2 stub; everything around them is reference code:
1. function level: rays_to_cameras_and_fixed_per_frame_intrinsics (geometry_utils.py:493-579) on synthetic ray maps of a known
   camera + noise vs oracle.rays_to_cameras_fixed_intrinsics (asserted <= 1e-5; the known K is recovered);
2. flow level: the reference's L4P_VideoMAE.forward on the mini geometry, 32 frames = 3 windows, tasks depth + camray, camray
   head as shipped (K estimated on the first window, reported for the later ones, whose rotations use the input K;
   dense_heads.py:303-334), joint alignment with its two random draws replaced as in tools/gen_golden_joint.py.  The oracle
   (OracleModel(use_intrinsics=False, seam="fixed")) must reproduce it (asserted <= 1e-4).
Only data is written: sampled outputs and the report."""
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
from oracle import l4p_oracle as lo
from tests.golden_utils import make_batch, sample_indices, synthetic_rays
from tools.gen_golden import build_reference, install_stubs, rel_err


def main():
    install_stubs()
    cv2 = sys.modules["cv2"]
    cv2.RANSAC = 8
    cv2.findHomography = lo.dlt_homography
    cv2.RQDecomp3x3 = lo.rq3
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    from l4p.utils import geometry_utils as gu
    import l4p.models.aligner as ref_al

    report = {}
    # ---- 1. function level ---------------------------------------------------------------------------------------
    rays, Ks = synthetic_rays()
    E_ref, _, K_ref = gu.rays_to_cameras_and_fixed_per_frame_intrinsics(rays.clone(), reproj_threshold=0.2, output_size=(224, 224))
    E_o, K_o = lo.rays_to_cameras_fixed_intrinsics(rays.clone(), (224, 224))
    report["fixed_intrinsics_extrinsics_rel_err"] = rel_err(E_o, E_ref)
    report["fixed_intrinsics_K_rel_err"] = rel_err(K_o, K_ref)
    assert report["fixed_intrinsics_extrinsics_rel_err"] <= 1e-5 and report["fixed_intrinsics_K_rel_err"] <= 1e-5, report
    # the known camera is recovered (ray-grid units): sanity of the stand-ins themselves
    K_ray = lo.denormalize_intrinsics(lo.normalize_intrinsics(K_ref, 224, 224), 16, 16)
    for b, K in enumerate(Ks):
        e = float((K_ray[b, :3, :3, 0] - K).abs().max() / K.abs().max())
        report[f"synthetic_K_recovery_rel_err_b{b}"] = e
        assert e <= 2e-2, (b, e, K_ray[b, :3, :3, 0], K)
    # the per-frame VARIABLE intrinsics branch (fixed_intrinsics=False, geometry_utils.py:582-654): every frame's own (R, K)
    Ev_ref, _, Kv_ref = gu.rays_to_cameras_and_variable_per_frame_intrinsics(rays.clone(), reproj_threshold=0.2, output_size=(224, 224))
    Ev_o, Kv_o = lo.rays_to_cameras_variable_intrinsics(rays.clone(), (224, 224))
    report["variable_intrinsics_extrinsics_rel_err"] = rel_err(Ev_o, Ev_ref)
    report["variable_intrinsics_K_rel_err"] = rel_err(Kv_o, Kv_ref)
    assert report["variable_intrinsics_extrinsics_rel_err"] <= 1e-5 and report["variable_intrinsics_K_rel_err"] <= 1e-5, report
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "intrinsics_variable.npz"), E=Ev_ref.numpy(), K=Kv_ref.numpy())
    # a supplied K (k_override) gives the same downstream result as estimating that K
    E_k, K_k = lo.rays_to_cameras_fixed_intrinsics(rays.clone(), (224, 224), k_override=lambda b: K_ray[b, :3, :3, 0])
    assert rel_err(E_k, E_ref) <= 1e-5 and rel_err(K_k, K_ref) <= 1e-5

    # ---- 2. the 3-window flow of the shipped configuration through the reference -----------------------------------
    cfg = ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    model = build_reference(cfg)
    model.load_state_dict(sd, strict=True)
    model.task_heads["camray"].use_intrinsics = False  # configs/model.yaml:44 (fixed_intrinsics: true is set at construction)
    assert model.task_heads["camray"].fixed_intrinsics
    batch = make_batch(32, 4)
    orig_est, orig_perm = ref_al.get_similarity_3d_transform, np.random.permutation
    ref_al.get_similarity_3d_transform = jo.fixed_inlier_estimator
    np.random.permutation = jo.fixed_permutation
    try:
        with torch.no_grad():
            out = model.forward({k: v.clone() for k, v in batch.items()}, ["depth", "camray"])
    finally:
        ref_al.get_similarity_3d_transform, np.random.permutation = orig_est, orig_perm
    om = lo.OracleModel(sd, cfg, use_intrinsics=False, seam="fixed")
    with torch.no_grad():
        oout = om.forward(batch, ["depth", "camray"])
    npz = {}
    for k in ("depth_est_b1thw", "traj3d_est_b16t", "traj3d_intrinsics_est_b16t"):
        e = rel_err(oout[k], out[k])
        report[f"flow_{k}_rel_err"] = e
        assert e <= 1e-4, (k, e)
        v = out[k].detach().float()
        npz[k] = v.reshape(-1)[sample_indices(v.numel())].numpy() if v.numel() > 4096 else v.numpy()
    # the estimate is the first window's and is reported for every frame of the clip
    Kc = out["traj3d_intrinsics_est_b16t"].reshape(1, 4, 4, -1)
    assert float((Kc - Kc[..., :1]).abs().max()) <= 1e-5 * float(Kc.abs().max())
    out_dir = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(out_dir, "mini_T32_default_config.npz"), **npz)
    with open(os.path.join(out_dir, "oracle_vs_reference_intrinsics.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
