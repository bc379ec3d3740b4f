"""ORACLE — test infrastructure only (see oracle/l4p_oracle.py for the rules: only tests/, smoke() and bench.py's
cpu_baseline leg may import this; the product path never does).

CPU restatement of the multi-window JOINT depth + camera estimation of NVlabs/L4P:

  joint_windowed_estimation     l4p/models/task_heads/dense_heads.py:360-492
  KabaschUmeyama3DAligner       l4p/models/aligner.py:158-265   (solve :177-237, apply :239-265)
  get_similarity_3d_transform   l4p/models/aligner.py:121-155
  generate_point_map            l4p/utils/geometry_utils.py:13-53

Everything in that path is deterministic reference arithmetic EXCEPT two draws: the 10 % point subset
(np.random.permutation, aligner.py:216-220) and skimage.measure.ransac's minimal samples (aligner.py:139-146).
scikit-image is a third-party dependency that is neither vendored in /root/reference nor listed in its
env/requirements.txt (no pinned version) and is not installed here; its two pieces are restated from their published
algorithms: SimilarityTransform.estimate = the closed-form least-squares similarity of Umeyama (IEEE PAMI 13(4), 1991,
eqs. 34-43) and ransac = "most inliers, ties by smaller residual sum, then re-estimate on the inliers of the best model".

Pinning (tools/gen_golden_joint.py, run where /root/reference is importable; fixture tests/golden/mini_T32_joint.npz):
  * generate_point_map, the q98 threshold and aligner.apply are compared with the IMPORTED reference functions;
  * the whole 3-window flow is run through the reference's own joint_windowed_estimation with the two draws replaced by
    the fixed stand-ins below (fixed_permutation / fixed_inlier_estimator, injected by monkeypatching the two names the
    reference calls) and must equal `joint_windowed(..., seam="fixed")` here.
The estimator the ENGINE runs (its counter-based sampler and trial schedule, csrc/umeyama.hip) is restated in
`engine_*` below so that a whole GPU solve is reproducible on the CPU (tests/test_joint_gpu.py); that estimator itself
is validated against synthetic ground truth (tests/test_umeyama_gpu.py) — the random draws of the reference cannot be
pinned by construction.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor

FRAME_SAMPLE_STEP = 3        # aligner.py:175
POINT_SAMPLE_RATIO = 0.1     # aligner.py:176
MIN_SAMPLES = 10             # aligner.py:172
REPROJ_THRESHOLD = 0.01      # aligner.py:173 (relative to the 98 % depth quantile, :187-188)
MAX_TRIALS = 100             # aligner.py:146


# --------------------------------------------------------------------------------------------------
# deterministic reference arithmetic
# --------------------------------------------------------------------------------------------------
def generate_point_map(depth_b1thw: Tensor, intrinsics_b44t: Tensor, world_T_cam_b44t: Tensor) -> Tensor:
    """geometry_utils.py:13-53: X_world = world_T_cam [depth * K^-1 (x, y, 1)^T ; 1], pixel (x = column, y = row)."""
    B, _, T, H, W = depth_b1thw.shape
    dt = depth_b1thw.dtype
    y, x = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pix = torch.stack([x, y, torch.ones_like(x)], dim=0)                                  # [3,H,W]
    Kinv = torch.inverse(intrinsics_b44t[:, :3, :3].permute(0, 3, 1, 2).float())          # [B,T,3,3]
    rays = torch.einsum("btmn,nhw->bmthw", Kinv, pix)                                     # [B,3,T,H,W]
    cam = rays * depth_b1thw
    cam4 = torch.cat([cam, torch.ones_like(cam[:, :1])], dim=1)
    return torch.einsum("bmnt,bnthw->bmthw", world_T_cam_b44t, cam4)[:, :3].to(dt)


def depth_q98(depth_b1thw: Tensor) -> Tensor:
    """aligner.py:187: torch.quantile(depth.reshape(bs, -1).float(), 0.98, dim=-1) (linear interpolation)."""
    return torch.quantile(depth_b1thw.reshape(depth_b1thw.shape[0], -1).float(), 0.98, dim=-1)


def umeyama(src_n3: np.ndarray, dst_n3: np.ndarray) -> Dict[str, np.ndarray]:
    """Least-squares similarity dst ~ s R src + t (Umeyama 1991): SVD of the cross-covariance, reflection fixed through
    the sign vector, s = trace(D S) / var(src).  float64.  Returns the dict get_similarity_3d_transform builds
    (aligner.py:148-153): T = [sR | t], s, t, and R = rotation / s as the reference (mis)labels it — only T and s are used."""
    src = np.asarray(src_n3, dtype=np.float64)
    dst = np.asarray(dst_n3, dtype=np.float64)
    n = src.shape[0]
    ms, md = src.mean(axis=0), dst.mean(axis=0)
    sc, dc = src - ms, dst - md
    cov = dc.T @ sc / n
    U, S, Vt = np.linalg.svd(cov)
    sign = np.ones(3)
    if np.linalg.det(cov) < 0:
        sign[2] = -1.0
    rank = np.linalg.matrix_rank(cov)
    if rank == 0:
        raise ValueError("degenerate point set")
    if rank == 2 and np.linalg.det(U) * np.linalg.det(Vt) < 0:
        R = U @ np.diag([1.0, 1.0, -1.0]) @ Vt
    elif rank == 2:
        R = U @ Vt
    else:
        R = U @ np.diag(sign) @ Vt
    s = float((S * sign).sum() / sc.var(axis=0).sum())
    t = md - s * (R @ ms)
    T = np.eye(4)
    T[:3, :3] = s * R
    T[:3, 3] = t
    return {"T": T, "R": R / s, "t": t, "s": np.float64(s)}


def similarity_apply(rel: Dict[str, Tensor], cur: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """KabaschUmeyama3DAligner.apply aligner.py:239-265: pose <- T pose, its 3x3 block / s; depth *= s; K untouched."""
    out = {}
    for k, v in cur.items():
        if k == "camray":
            bs, _, T = v.shape
            pose = torch.einsum("bij,bjkt->bikt", rel["T"], v.reshape(bs, 4, 4, T)).clone()
            pose[:, :3, :3] = pose[:, :3, :3] / rel["s"]
            out[k] = pose.reshape(bs, -1, T)
        elif k == "depth":
            out[k] = v * rel["s"]
        elif k == "camray_intrinsics_est":
            out[k] = v
        else:
            raise ValueError(f"Unknown task name: {k}")
    return out


# --------------------------------------------------------------------------------------------------
# fixed stand-ins for the two random draws (used to pin the FLOW against the reference, see module docstring)
# --------------------------------------------------------------------------------------------------
def fixed_permutation(n: int) -> np.ndarray:
    """A fixed permutation of range(n) with no RNG behind it: i -> (a i + 13) mod n, a = first integer >= 7919 coprime to n."""
    a = 7919
    while np.gcd(a, n) != 1:
        a += 1
    return (np.arange(n, dtype=np.int64) * a + 13) % n


def fixed_inlier_estimator(src_n3, dst_n3, min_samples=None, reprojection_threshold=None, confidence=None):
    """Stand-in for get_similarity_3d_transform with the RANSAC draw removed: closed-form Umeyama on a FIXED inlier set
    (every correspondence whose index is not a multiple of 3).  Same signature and return value as aligner.py:121-155."""
    inl = (np.arange(src_n3.shape[0]) % 3) != 0
    return umeyama(src_n3[inl], dst_n3[inl]), inl


# --------------------------------------------------------------------------------------------------
# the engine's deterministic estimator (csrc/umeyama.hip), restated
# --------------------------------------------------------------------------------------------------
ENGINE_SEED = 20250213  # l4p_amd/utils/umeyama.py


def hash_u32(x) -> np.ndarray:
    """csrc/umeyama.hip:hash_u32 (PCG-style integer hash) on uint32 arrays."""
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    w = (((x >> ((x >> 28) + 4)) ^ x) * 277803737) & 0xFFFFFFFF
    return ((w >> 22) ^ w).astype(np.uint64)


def engine_pixel_subset(H: int, W: int, ratio: int, seed: int = ENGINE_SEED) -> np.ndarray:
    """pointmap_kernel: sample j of a frame is pixel j*ratio + hash(seed ^ j*2654435761) % ratio (one per stride cell)."""
    j = np.arange((H * W) // ratio, dtype=np.uint64)
    pix = j * ratio + hash_u32((np.uint64(seed) ^ (j * 2654435761)) & 0xFFFFFFFF) % np.uint64(ratio)
    return np.minimum(pix, H * W - 1).astype(np.int64)


def _residuals(T: np.ndarray, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    m = T.astype(np.float32)
    r = src @ m[:3, :3].T + m[:3, 3] - dst
    return np.sqrt((r * r).sum(axis=1, dtype=np.float32))


def engine_ransac(src_n3: np.ndarray, dst_n3: np.ndarray, thr: float, trials: int = MAX_TRIALS,
                  min_samples: int = MIN_SAMPLES, seed: int = ENGINE_SEED) -> Tuple[Dict[str, np.ndarray], np.ndarray]:
    """ransac_trials_kernel + ransac_final_kernel: trial t draws points hash(seed + 7919 t + 104729 j) % n, j < min_samples;
    best = most inliers (residual < thr), ties by the smaller residual sum; the best model is re-estimated on its inliers
    (kept as is when fewer than 3)."""
    src = np.asarray(src_n3, dtype=np.float32)
    dst = np.asarray(dst_n3, dtype=np.float32)
    n = src.shape[0]
    thr = np.float32(thr)
    best = None
    for t in range(trials):
        idx = (hash_u32((seed + 7919 * t + 104729 * np.arange(min_samples, dtype=np.uint64)) & 0xFFFFFFFF) % np.uint64(n)).astype(np.int64)
        model = umeyama(src[idx], dst[idx])
        r = _residuals(model["T"], src, dst)
        inl = r < thr
        score = (int(inl.sum()), -float(r[inl].sum(dtype=np.float64)))
        if best is None or score > best[0]:
            best = (score, model, inl)
    _, model, inl = best
    if int(inl.sum()) >= 3:
        model = umeyama(src[inl], dst[inl])
    return model, inl


# --------------------------------------------------------------------------------------------------
# the seam solve and the window loop
# --------------------------------------------------------------------------------------------------
def seam_solve(pred: Dict[str, Tensor], target: Dict[str, Tensor], seam: str, log: Optional[list] = None) -> Dict[str, Tensor]:
    """KabaschUmeyama3DAligner.solve aligner.py:177-237 with the draws chosen by ``seam``:
    "fixed"  - fixed_permutation + fixed_inlier_estimator (the flow pinned against the reference),
    "engine" - the engine's hashed pixel subset + trial schedule (what libl4p_hip.so computes)."""
    bs, _, ov, H, W = pred["depth"].shape
    q98 = depth_q98(pred["depth"])
    thr = (q98 * REPROJ_THRESHOLD).numpy()
    step = FRAME_SAMPLE_STEP
    pm = {}
    for name, d in (("pred", pred), ("target", target)):
        pm[name] = generate_point_map(d["depth"][:, :, ::step], d["camray_intrinsics"].reshape(bs, 4, 4, -1)[..., ::step],
                                      d["camray"].reshape(bs, 4, 4, -1)[..., ::step])
    rels = []
    for b in range(bs):
        xp = pm["pred"][b].reshape(3, -1).numpy().T.astype(np.float32)
        xt = pm["target"][b].reshape(3, -1).numpy().T.astype(np.float32)
        if seam == "fixed":
            n = xp.shape[0]
            idx = fixed_permutation(n)[: int(POINT_SAMPLE_RATIO * n)]
            rel, inl = fixed_inlier_estimator(xp[idx], xt[idx], MIN_SAMPLES, thr[b], 0.99)
        elif seam == "engine":
            ratio = int(round(1.0 / POINT_SAMPLE_RATIO))
            F = pm["pred"].shape[2]
            sub = engine_pixel_subset(H, W, ratio)
            idx = (np.arange(F)[:, None] * (H * W) + sub[None, :]).reshape(-1)
            rel, inl = engine_ransac(xp[idx], xt[idx], thr[b])
        else:
            raise ValueError(seam)
        if log is not None:
            log.append({"thr": float(thr[b]), "T": rel["T"].copy(), "s": float(rel["s"]), "inliers": int(inl.sum()), "n": len(idx)})
        rels.append(rel)
    dt = pred["depth"].dtype
    return {k: torch.from_numpy(np.stack([np.asarray(r[k]) for r in rels], axis=0)).to(dt) for k in rels[0]}


def joint_windowed(heads: Callable[[int], Dict[str, Tensor]], strides: Sequence[int], ws: int, seam: str,
                   log: Optional[list] = None) -> Dict[str, Tensor]:
    """joint_windowed_estimation dense_heads.py:360-492 for time_strides with >= 1 entries.  ``heads(win_id)`` returns
    the per-window estimates {"depth": [B,1,ws,H,W], "camray": [B,16,ws], "camray_intrinsics_est": [B,16,ws]} (the
    reference's per-window head forwards, :404-422)."""
    T = int(strides[-1]) + ws
    est: Dict[str, Optional[Tensor]] = {}
    for win_id, st in enumerate(int(s) for s in strides):
        cur = heads(win_id)
        for k, v in cur.items():
            if k not in est:
                shp = list(v.shape)
                shp[2] = T
                est[k] = torch.zeros(*shp, dtype=v.dtype)
        if win_id > 0:
            ov = int(strides[win_id - 1]) + ws - st
            pred = {"depth": cur["depth"][:, :, :ov], "camray": cur["camray"][:, :, :ov],
                    "camray_intrinsics": cur["camray_intrinsics_est"][:, :, :ov].reshape(-1, 4, 4, ov).clone()}
            target = {"depth": est["depth"][:, :, st:st + ov], "camray": est["camray"][:, :, st:st + ov],
                      "camray_intrinsics": est["camray_intrinsics_est"][:, :, st:st + ov].reshape(-1, 4, 4, ov)}
            rel = seam_solve(pred, target, seam, log)
            cur = similarity_apply(rel, cur)
        for k, v in cur.items():
            est[k][:, :, st:st + ws] = v
    return est
