"""ORACLE — test infrastructure only.

A plain-PyTorch fp32 CPU restatement of the NVlabs/L4P inference algorithm (shared VideoMAE-v2
encoder + dense DPT heads + ray/pose head + SAM-style tracker + window stitching), written
functionally over a reference-format state_dict.  Every function cites the reference file:line it
follows.  It is pinned against the real reference by tools/gen_golden.py (run in the build container,
where /root/reference is importable) through the fixtures committed under tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (l4p_amd/*) never does and fails loudly when the HIP library is missing.

Parity status: encoder, DPT heads, LstSq window stitching, tracker (incl. multi-window memory /
re-seeding), rays->camera with given intrinsics and the multi-window joint depth + camera flow
(oracle/joint_oracle.py: point maps, q98 threshold, similarity apply, stitching) are pinned by golden
vectors, and so is the K-estimation flow of the shipped default (use_intrinsics=false) AROUND its two cv2
calls (tools/gen_golden_intrinsics.py: the reference run with deterministic stand-ins in its cv2 stub).
The RANDOM DRAWS of the two third-party RANSAC steps of the reference (cv2.findHomography, skimage.measure.
ransac — unpinned versions, not installed here) are "parity unpinned": see DESIGN.md section 7.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------------------
def sinusoid_table(n_position: int, d_hid: int) -> Tensor:
    """modeling_finetune.py:288-299 — float64 numpy table, cast to float32 at the end."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid, dtype=np.float64)[None, :]
    ang = pos / np.power(10000.0, 2.0 * np.floor(j / 2.0) / d_hid)
    tab = np.empty_like(ang)
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.tensor(tab, dtype=torch.float32).unsqueeze(0)


def encoder_block(sd: Dict[str, Tensor], p: str, x: Tensor, heads: int, eps: float) -> Tensor:
    """Block.forward modeling_finetune.py:245-252 with Attention :169-190 and Mlp :62-69 (gamma_* = None)."""
    B, N, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    qkv_bias = torch.cat((sd[p + "attn.q_bias"], torch.zeros_like(sd[p + "attn.v_bias"]), sd[p + "attn.v_bias"]))
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], qkv_bias).reshape(B, N, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    scale = (C // heads) ** -0.5
    attn = ((q * scale) @ k.transpose(-2, -1)).softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def encoder_forward(sd: Dict[str, Tensor], rgb: Tensor, cfg, upto: Optional[int] = None) -> List[Tensor]:
    """VideoMAEEncoder.forward l4p_videomae.py:80-122: returns [embeddings, after block 1, ..., norm(last)].
    ``upto`` truncates the block loop (testing aid); the final norm is applied only at full depth."""
    p = "video_encoder."
    x = F.conv3d(rgb, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=tuple(cfg.patch))
    x = x.flatten(2).transpose(1, 2)  # modeling_finetune.py:282
    x = x + sinusoid_table(x.shape[1], cfg.dim)
    feats = [x]
    depth = cfg.depth if upto is None else upto
    for i in range(depth):
        feats.append(encoder_block(sd, f"{p}blocks.{i}.", feats[-1], cfg.heads, cfg.ln_eps))
    if depth == cfg.depth:
        feats[-1] = F.layer_norm(feats[-1], (cfg.dim,), sd[p + "norm.weight"], sd[p + "norm.bias"], cfg.ln_eps)
    return feats


# --------------------------------------------------------------------------------------------------
# DPT decoder
# --------------------------------------------------------------------------------------------------
def _rcu(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """ResidualConvUnit_custom.forward dpt_block.py:131-157 (bn=False, ReLU pre-activations)."""
    out = F.conv3d(F.relu(x), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    out = F.conv3d(F.relu(out), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    return out + x


def _fusion(sd: Dict[str, Tensor], p: str, scale: Sequence[int], x0: Tensor, x1: Optional[Tensor] = None) -> Tensor:
    """FeatureFusionBlock_custom.forward dpt_block.py:210-238."""
    out = x0
    if x1 is not None:
        out = out + _rcu(sd, p + "resConfUnit1.", x1)
    out = _rcu(sd, p + "resConfUnit2.", out)
    out = F.interpolate(out, scale_factor=tuple(float(s) for s in scale), mode="trilinear", align_corners=True)
    return F.conv3d(out, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


def _actpost(sd: Dict[str, Tensor], p: str, x: Tensor, sf: Sequence[int]) -> Tensor:
    """act_postprocess[i] = Conv3d 1x1x1 + make_conv3d_custom  (dpt_block.py:255-278, :447-507)."""
    x = F.conv3d(x, sd[p + "0.weight"], sd[p + "0.bias"])
    if any(s > 0 for s in sf):
        st = tuple(2 ** s for s in sf)
        x = F.conv_transpose3d(x, sd[p + "1.weight"], sd[p + "1.bias"], stride=st)
    elif any(s < 0 for s in sf):
        st = tuple(2 ** (-s) for s in sf)
        pad = tuple(s // 2 for s in st)
        x = F.conv3d(x, sd[p + "1.weight"], sd[p + "1.bias"], stride=st, padding=pad)
    return x


def dpt_forward(sd: Dict[str, Tensor], task: str, feats: Sequence[Tensor], cfg, actpost, fusion,
                output_size: Optional[Tuple[int, int, int]], image_size=(16, 224, 224)) -> Tensor:
    """DPTOutputAdapter_fix.forward dpt_head.py:41-86.  feats is the encoder feature list."""
    p = f"task_heads.{task}.task_head.dpt."
    nt, nh, nw = cfg.grid
    layers = []
    for i, hook in enumerate(cfg.hooks):
        tok = feats[hook]
        B = tok.shape[0]
        x = tok.reshape(B, nt, nh, nw, cfg.dim).permute(0, 4, 1, 2, 3).contiguous()  # b (nt nh nw) c -> b c nt nh nw
        x = _actpost(sd, f"{p}act_postprocess.{i}.", x, actpost[i])
        layers.append(F.conv3d(x, sd[f"{p}scratch.layer_rn.{i}.weight"], None, padding=1))
    s = p + "scratch."
    path4 = _fusion(sd, s + "refinenet4.", fusion[3], layers[3])[:, :, : layers[2].shape[2], : layers[2].shape[3]]
    path3 = _fusion(sd, s + "refinenet3.", fusion[2], path4, layers[2])
    path2 = _fusion(sd, s + "refinenet2.", fusion[1], path3, layers[1])
    path1 = _fusion(sd, s + "refinenet1.", fusion[0], path2, layers[0])
    out = F.conv3d(path1, sd[p + "head1.0.weight"], sd[p + "head1.0.bias"], padding=1)
    osz = tuple(image_size) if output_size is None else tuple(output_size)
    if tuple(out.shape[-3:]) != osz:
        out = F.interpolate(out, size=osz, mode="trilinear", align_corners=True)
    out = F.relu(F.conv3d(out, sd[p + "head2.0.weight"], sd[p + "head2.0.bias"], padding=1))
    return F.conv3d(out, sd[p + "head2.2.weight"], sd[p + "head2.2.bias"])


# --------------------------------------------------------------------------------------------------
# small utilities (l4p/utils/misc.py, l4p/utils/geometry_utils.py)
# --------------------------------------------------------------------------------------------------
def safe_inverse(x: Tensor, keep_above: float = 0.0) -> Tensor:
    """misc.py:48-62."""
    out = torch.zeros_like(x)
    m = x > keep_above
    out[m] = 1.0 / x[m]
    return out


def normalize_intrinsics(K_b44t: Tensor, h: int, w: int) -> Tensor:
    """geometry_utils.py:110-116."""
    K = K_b44t.clone()
    K[:, :2, 2] += 0.5
    K[:, 0] = K[:, 0] / w
    K[:, 1] = K[:, 1] / h
    return K


def denormalize_intrinsics(K_b44t: Tensor, h: int, w: int) -> Tensor:
    """geometry_utils.py:119-125."""
    K = K_b44t.clone()
    K[:, 0] *= w
    K[:, 1] *= h
    K[:, :2, 2] -= 0.5
    return K


def lstsq_affine_solve(pred: Tensor, target: Tensor) -> Tensor:
    """LstSqAffineAligner.solve with pre_post_fn='inverse' (aligner.py:45-56): scale/shift in 1/depth."""
    a = safe_inverse(pred).reshape(pred.shape[0], -1, 1).float()
    b = safe_inverse(target).reshape(target.shape[0], -1, 1).float()
    A = torch.cat([a, torch.ones_like(a)], dim=-1)
    return torch.linalg.lstsq(A, b, rcond=None).solution[..., 0]  # B x 2


def lstsq_affine_apply(pred: Tensor, sol: Tensor) -> Tensor:
    """LstSqAffineAligner.apply (aligner.py:58-66)."""
    shp = (sol.shape[0],) + (1,) * (pred.ndim - 1)
    return safe_inverse(sol[:, 0].reshape(shp) * safe_inverse(pred) + sol[:, 1].reshape(shp))


def linear_mean_solve(pred: Tensor, target: Tensor) -> Tensor:
    """LinearAligner(pre_post_fn='inverse', method='mean').solve (aligner.py:91-109): mean of 1/target over (1/pred + 1e-8)."""
    a = safe_inverse(pred).reshape(pred.shape[0], -1)
    b = safe_inverse(target).reshape(target.shape[0], -1)
    return torch.mean(b / (a + 1e-8), dim=1)


def linear_median_solve(pred: Tensor, target: Tensor, inverse: bool = True) -> Tensor:
    """LinearAligner(method='median').solve (aligner.py:91-109): torch.median (the LOWER median) of the float ratios."""
    f = safe_inverse if inverse else (lambda x: x)
    a = f(pred).reshape(pred.shape[0], -1)
    b = f(target).reshape(target.shape[0], -1)
    return torch.median(b / (a + 1e-8), dim=1).values


def linear_mean_apply(pred: Tensor, scale: Tensor) -> Tensor:
    """LinearAligner.apply (aligner.py:111-118)."""
    return safe_inverse(scale.reshape((scale.shape[0],) + (1,) * (pred.ndim - 1)) * safe_inverse(pred))


def plucker_to_point_direction(rays_b6thw: Tensor) -> Tuple[Tensor, Tensor]:
    """geometry_utils.py:308-328."""
    d = rays_b6thw[:, :3]
    m = rays_b6thw[:, 3:] / torch.linalg.norm(d, dim=1, keepdim=True)
    return torch.cross(d, m, dim=1), d


def intersect_skew_lines(points: Tensor, directions: Tensor) -> Tensor:
    """geometry_utils.py:249-282 (mask = ones)."""
    d = F.normalize(directions, dim=-1)
    eye = torch.eye(3, dtype=points.dtype)[None, None]
    P = eye - d[..., None] * d[..., None, :]
    rhs = P.matmul(points[..., None]).sum(dim=-3)
    return torch.linalg.lstsq(P.float().sum(dim=-3), rhs.float()).solution[..., 0]


def kabsch_rotation(A: Tensor, Bm: Tensor) -> Tensor:
    """compute_optimal_rotation_alignment geometry_utils.py:285-305: argmin_R ||A - B R||_F."""
    Hm = (Bm.T @ A).float()
    U, _, Vh = torch.linalg.svd(Hm, full_matrices=True)
    s = torch.linalg.det(U @ Vh)
    Sp = torch.diag(torch.tensor([1.0, 1.0, float(torch.sign(s))]))
    return (U @ Sp @ Vh).T


def rays_to_cameras(rays_b6thw: Tensor, Kn_b44t: Tensor) -> Tensor:
    """geometry_utils.py:331-406 with ctr_only=False: extrinsics (cam_T_world) from a Pluecker ray map
    and NORMALISED intrinsics."""
    B, _, T, h, w = rays_b6thw.shape
    rays = rays_b6thw.to(Kn_b44t.dtype)
    origins, directions = plucker_to_point_direction(rays)
    o = origins.permute(0, 2, 3, 4, 1).reshape(-1, h * w, 3)
    d = directions.permute(0, 2, 3, 4, 1).reshape(-1, h * w, 3)
    centers = intersect_skew_lines(o, d).reshape(B, T, 3)
    K = denormalize_intrinsics(Kn_b44t, h, w)[:, :3, :3]
    j, i = torch.meshgrid(torch.arange(h, dtype=rays.dtype), torch.arange(w, dtype=rays.dtype), indexing="ij")
    pix = torch.stack([i.expand(B, -1, -1), j.expand(B, -1, -1), torch.ones_like(i).expand(B, -1, -1)], dim=-1)
    rd = torch.einsum("btmn,bhwn->bthwm", torch.inverse(K.permute(0, 3, 1, 2)), pix)
    rd = rd / rd.norm(dim=-1, keepdim=True)
    E = torch.zeros_like(Kn_b44t)
    E[:, 3, 3] = 1.0
    for b in range(B):
        for t in range(T):
            E[b, :3, :3, t] = kabsch_rotation(rd[b, t].reshape(-1, 3), directions[b, :, t].reshape(3, -1).T)
    tr = -torch.matmul(E[:, :3, :3].permute(0, 3, 1, 2), centers[..., None]).squeeze(3)
    E[:, :3, -1] = tr.permute(0, 2, 1)
    return E


# ---- K estimation (shipped default: use_intrinsics=false, fixed_intrinsics=true) ------------------------------------
# The reference estimates K from the first frame's ray map with cv2.findHomography(RANSAC) + cv2.RQDecomp3x3
# (geometry_utils.py:436-448; opencv-python unpinned in env/requirements.txt, not installed here: the RANSAC draw cannot be
# pinned).  Everything AROUND those two calls is restated here with the two calls injectable, and pinned against the imported
# reference with the same stand-ins installed in its cv2 stub (tools/gen_golden_intrinsics.py).
def dlt_homography(src_n2: np.ndarray, dst_n2: np.ndarray, method=None, reproj_threshold=None):
    """Deterministic stand-in for cv2.findHomography(src, dst, cv2.RANSAC, thr): Hartley-normalised DLT over ALL
    correspondences in float64 (no consensus step).  Returns (H 3x3 float64 with H[2,2] = 1, None) like cv2."""
    src, dst = np.asarray(src_n2, np.float64), np.asarray(dst_n2, np.float64)

    def norm(p):
        c = p.mean(0)
        s = np.sqrt(2.0) / max(np.sqrt(((p - c) ** 2).sum(1)).mean(), 1e-12)
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
        return (p - c) * s, T

    a, Ta = norm(src)
    b, Tb = norm(dst)
    n = a.shape[0]
    A = np.zeros((2 * n, 9))
    A[0::2, 0:2], A[0::2, 2] = a, 1.0
    A[0::2, 6:8], A[0::2, 8] = -b[:, :1] * a, -b[:, 0]
    A[1::2, 3:5], A[1::2, 5] = a, 1.0
    A[1::2, 6:8], A[1::2, 8] = -b[:, 1:2] * a, -b[:, 1]
    _, _, Vt = np.linalg.svd(A, full_matrices=False)
    Hn = Vt[-1].reshape(3, 3)
    H = np.linalg.inv(Tb) @ Hn @ Ta
    return H / H[2, 2], None


def rq3(H: np.ndarray):
    """Deterministic stand-in for cv2.RQDecomp3x3(H): H = K R with K upper triangular (positive diagonal) and R orthogonal.
    Returns a tuple indexed like cv2's: [1] = K, [2] = R."""
    H = np.asarray(H, np.float64)
    P = np.eye(3)[::-1]
    q, r = np.linalg.qr((P @ H).T)
    K, R = P @ r.T @ P, P @ q.T
    S = np.diag(np.sign(np.diag(K)) + (np.diag(K) == 0))
    K, R = K @ S, S @ R
    return None, K, R


def optimal_rotation_intrinsics(rays_origin: Tensor, rays_target: Tensor, find_h=dlt_homography, rq=rq3,
                                z_threshold: float = 1e-4, reproj_threshold: float = 0.2):
    """compute_optimal_rotation_intrinsics geometry_utils.py:409-456 with the two cv2 calls injected."""
    z_mask = torch.logical_and(torch.abs(rays_target) > z_threshold, torch.abs(rays_origin) > z_threshold)[:, 2]
    rt, ro = rays_target[z_mask], rays_origin[z_mask]
    ro = ro[:, :2] / ro[:, -1:]
    rt = rt[:, :2] / rt[:, -1:]
    A, _ = find_h(ro.numpy(), rt.numpy(), None, reproj_threshold)
    A = torch.from_numpy(np.asarray(A)).float()
    if torch.linalg.det(A) < 0:
        A = -A
    Hm = torch.linalg.inv(A.float())  # H = K @ R
    out = rq(Hm.numpy())
    K = np.asarray(out[1])
    K = K / K[2, 2]
    return torch.from_numpy(np.asarray(out[2])).float(), torch.from_numpy(K).float(), Hm


def rays_to_cameras_variable_intrinsics(rays_b6thw: Tensor, output_size: Tuple[int, int], find_h=dlt_homography, rq=rq3,
                                        estimate=None) -> Tuple[Tensor, Tensor]:
    """rays_to_cameras_and_variable_per_frame_intrinsics geometry_utils.py:582-654 (ctr_only=False; reached from
    dense_heads.py:336-344 when fixed_intrinsics=False): for EVERY frame, (R, K) = compute_optimal_rotation_intrinsics of the
    identity-intrinsics rays against that frame's predicted directions - R is the rotation of the RQ step itself, no Kabsch -,
    translation -R c from the ray intersection, K rescaled from the ray grid to ``output_size``.  ``estimate(b, t, rays_origin,
    rays_target)`` -> (R 3x3, K 3x3 ray-grid) replaces the two third-party calls by another estimator (the engine's).
    Returns (cam_T_world [B,4,4,T], K [B,4,4,T] in output pixels)."""
    B, _, T, h, w = rays_b6thw.shape
    rays = rays_b6thw.float()
    origins, directions = plucker_to_point_direction(rays)
    o = origins.permute(0, 2, 3, 4, 1).reshape(-1, h * w, 3)
    d = directions.permute(0, 2, 3, 4, 1).reshape(-1, h * w, 3)
    centers = intersect_skew_lines(o, d).reshape(B, T, 3)
    j, i = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    pix = torch.stack([i, j, torch.ones_like(i)], dim=-1).reshape(-1, 3)
    rd = pix / pix.norm(dim=-1, keepdim=True)  # rays of the identity intrinsics
    E = torch.zeros(B, 4, 4, T)
    E[:, 3, 3] = 1.0
    Kest = torch.zeros_like(E)
    Kest[:, 3, 3] = 1.0
    Kest[:, 2, 2] = 1.0
    for b in range(B):
        for t in range(T):
            tgt = directions[b, :, t].reshape(3, -1).T.float()
            if estimate is not None:
                R, K = estimate(b, t, rd, tgt)
            else:
                R, K, _ = optimal_rotation_intrinsics(rd, tgt, find_h, rq, reproj_threshold=0.2)
            E[b, :3, :3, t] = torch.as_tensor(R).float()
            Kest[b, :3, :3, t] = torch.as_tensor(K).float()
    tr = -torch.matmul(E[:, :3, :3].permute(0, 3, 1, 2), centers[..., None]).squeeze(3)
    E[:, :3, -1] = tr.permute(0, 2, 1)
    Ho, Wo = output_size
    return E, denormalize_intrinsics(normalize_intrinsics(Kest, h, w), Ho, Wo)


# ---- the ENGINE's deterministic K estimator (csrc/intrinsics.hip rays_to_intrinsics_kernel), restated --------------------------
# It stands where cv2's RANSAC draw stands in the reference, so it has no reference result to be compared with; restating
# its schedule here makes a GPU estimate reproducible on the CPU at any geometry (tests/test_full_model_gpu.py compares the
# full-size model's estimate with it), exactly as joint_oracle.engine_ransac does for the seam similarity.
def _engine_dlt(x: np.ndarray, y: np.ndarray, u: np.ndarray, v: np.ndarray) -> np.ndarray:
    """Hartley-normalised DLT of (x, y) -> (u, v) in float64: null vector of the 9x9 normal matrix, denormalised."""
    n = x.shape[0]
    mx, my, mu, mv = x.mean(), y.mean(), u.mean(), v.mean()
    ds = np.sqrt((x - mx) ** 2 + (y - my) ** 2).sum()
    dd = np.sqrt((u - mu) ** 2 + (v - mv) ** 2).sum()
    ss = np.sqrt(2.0) * n / ds if ds > 0 else 1.0
    sd = np.sqrt(2.0) * n / dd if dd > 0 else 1.0
    xn, yn, un, vn = (x - mx) * ss, (y - my) * ss, (u - mu) * sd, (v - mv) * sd
    z, o = np.zeros(n), np.ones(n)
    r1 = np.stack([-xn, -yn, -o, z, z, z, un * xn, un * yn, un], axis=1)
    r2 = np.stack([z, z, z, -xn, -yn, -o, vn * xn, vn * yn, vn], axis=1)
    M = r1.T @ r1 + r2.T @ r2
    _, V = np.linalg.eigh(M)
    Hn = V[:, 0].reshape(3, 3)
    Ts = np.array([[ss, 0, -ss * mx], [0, ss, -ss * my], [0, 0, 1.0]])
    Tdi = np.array([[1 / sd, 0, mu], [0, 1 / sd, mv], [0, 0, 1.0]])
    return Tdi @ Hn @ Ts


def _engine_reproj(Hm: np.ndarray, x, y, u, v, dt) -> np.ndarray:
    Hm = Hm.astype(dt)
    x, y, u, v = x.astype(dt), y.astype(dt), u.astype(dt), v.astype(dt)
    with np.errstate(all="ignore"):
        pw = Hm[2, 0] * x + Hm[2, 1] * y + Hm[2, 2]
        pu = (Hm[0, 0] * x + Hm[0, 1] * y + Hm[0, 2]) / pw
        pv = (Hm[1, 0] * x + Hm[1, 1] * y + Hm[1, 2]) / pw
        return np.sqrt((pu - u) ** 2 + (pv - v) ** 2)


def engine_rays_to_intrinsics(dirs_n3: np.ndarray, h: int, w: int, H: int, W: int, thr: float = 0.2, b: int = 0,
                              z_thr: float = 1e-4):
    """rays_to_intrinsics_kernel for batch item ``b``: dirs_n3 = predicted ray directions of the first frame on the h x w ray
    grid (row-major).  128 hashed minimal 4-point homographies (grid pixel -> direction xy / z) scored by consensus in float,
    the best one's consensus set re-estimated by DLT until the set is stable (<= 8 rounds, kept when a new set would have < 8
    members), H^-1 = K R by RQ with a positive diagonal, K rescaled from the ray grid to the H x W image.
    Returns (K 4x4 float64 in output pixels, consensus size, rounds)."""
    d = np.asarray(dirs_n3, np.float32).astype(np.float64)
    nr = h * w
    r = np.arange(nr)
    x, y = (r % w).astype(np.float64), (r // w).astype(np.float64)
    valid = np.abs(d[:, 2]) > z_thr
    u = np.where(valid, d[:, 0] / np.where(valid, d[:, 2], 1.0), 0.0)
    v = np.where(valid, d[:, 1] / np.where(valid, d[:, 2], 1.0), 0.0)
    f32 = np.float32
    xf, yf, uf, vf = x.astype(f32).astype(np.float64), y.astype(f32).astype(np.float64), u.astype(f32).astype(np.float64), v.astype(f32).astype(np.float64)
    counts, Hts = np.zeros(128, np.int64), []
    for t in range(128):
        idx, ok = [], True
        for k in range(4):
            hsh = ((b * 131 + t) * 2654435761 + 40503 * k) & 0xFFFFFFFF
            hsh ^= hsh >> 15
            hsh = (hsh * 2246822519) & 0xFFFFFFFF
            hsh ^= hsh >> 13
            i = hsh % nr
            ok = ok and bool(valid[i]) and i not in idx
            idx.append(i)
        if not ok:
            Hts.append(np.zeros((3, 3)))
            continue
        idx = np.array(idx)
        Ht = _engine_dlt(xf[idx], yf[idx], uf[idx], vf[idx])
        Hts.append(Ht)
        e = _engine_reproj(Ht, x, y, u, v, f32)
        counts[t] = int((valid & (e < f32(thr))).sum())
    bt = int(np.argmax(counts))  # first maximum, as the kernel's strict > scan
    wgt = valid & (_engine_reproj(Hts[bt], x, y, u, v, f32) < f32(thr))
    if counts[bt] < 8:
        wgt = valid.copy()
    Hs, iters = None, 0
    for _ in range(8):
        if int(wgt.sum()) < 4:
            break
        Hs = _engine_dlt(x[wgt], y[wgt], u[wgt], v[wgt])
        nw = valid & (_engine_reproj(Hs, x, y, u, v, np.float64) < thr)
        iters += 1
        if np.array_equal(nw, wgt) or int(nw.sum()) < 8:
            break
        wgt = nw
    A = Hs if np.linalg.det(Hs) >= 0 else -Hs
    _, K, Rm = rq3(np.linalg.inv(A))
    engine_rays_to_intrinsics.last_R = Rm  # rotation of H^-1 = K R (the per-frame variable branch uses it as the camera rotation)
    K = K / K[2, 2]
    K4 = np.eye(4)
    K4[:3, :3] = K
    K4[0, 2] += 0.5
    K4[1, 2] += 0.5
    K4[0, :] = K4[0, :] / w * W
    K4[1, :] = K4[1, :] / h * H
    K4[0, 2] -= 0.5
    K4[1, 2] -= 0.5
    return K4, int(wgt.sum()), iters


def rays_to_cameras_fixed_intrinsics(rays_b6thw: Tensor, output_size: Tuple[int, int], find_h=dlt_homography, rq=rq3,
                                     k_override=None) -> Tuple[Tensor, Tensor]:
    """rays_to_cameras_and_fixed_per_frame_intrinsics geometry_utils.py:493-579 (ctr_only=False): K from the FIRST frame's
    rays on the ray grid (identity-intrinsics rays -> predicted directions), repeated over the window; rotations by Kabsch
    against rays of that K; translation from the ray intersection; K rescaled to ``output_size``.  ``k_override(b)``
    supplies the 3x3 ray-grid K instead of estimating it (closed form on a supplied estimate: how the engine's own
    estimator is tied to everything downstream of it).  Returns (cam_T_world [B,4,4,T], K [B,4,4,T] in output pixels)."""
    B, _, T, h, w = rays_b6thw.shape
    rays = rays_b6thw.float()
    origins, directions = plucker_to_point_direction(rays)
    o = origins.permute(0, 2, 3, 4, 1).reshape(-1, h * w, 3)
    d = directions.permute(0, 2, 3, 4, 1).reshape(-1, h * w, 3)
    centers = intersect_skew_lines(o, d).reshape(B, T, 3)
    j, i = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    pix = torch.stack([i.expand(B, -1, -1), j.expand(B, -1, -1), torch.ones_like(i).expand(B, -1, -1)], dim=-1)
    E = torch.zeros(B, 4, 4, T)
    E[:, 3, 3] = 1.0
    Kest = torch.zeros_like(E)
    Kest[:, 3, 3] = 1.0
    Kest[:, 2, 2] = 1.0
    eye = torch.eye(3)[None, :, :, None].repeat(B, 1, 1, T)
    rd = torch.einsum("btmn,bhwn->bthwm", torch.inverse(eye.permute(0, 3, 1, 2)), pix)
    rd = rd / rd.norm(dim=-1, keepdim=True)
    for b in range(B):
        if k_override is not None:
            K = k_override(b).float()
        else:
            _, K, _ = optimal_rotation_intrinsics(rd[b, 0].reshape(-1, 3).float(), directions[b, :, 0].reshape(3, -1).T.float(),
                                                  find_h, rq, reproj_threshold=0.2)
        Kest[b, :3, :3, :] = K[:, :, None].repeat(1, 1, T)
    rd = torch.einsum("btmn,bhwn->bthwm", torch.inverse(Kest[:, :3, :3].permute(0, 3, 1, 2)), pix)
    rd = rd / rd.norm(dim=-1, keepdim=True)
    for b in range(B):
        for t in range(T):
            E[b, :3, :3, t] = kabsch_rotation(rd[b, t].reshape(-1, 3), directions[b, :, t].reshape(3, -1).T)
    tr = -torch.matmul(E[:, :3, :3].permute(0, 3, 1, 2), centers[..., None]).squeeze(3)
    E[:, :3, -1] = tr.permute(0, 2, 1)
    Ho, Wo = output_size
    return E, denormalize_intrinsics(normalize_intrinsics(Kest, h, w), Ho, Wo)


# --------------------------------------------------------------------------------------------------
# tracker (sparse_heads.py + sam/*)
# --------------------------------------------------------------------------------------------------
def _pe_encoding(G: Tensor, coords01: Tensor) -> Tensor:
    """PositionEmbeddingRandom3D._pe_encoding prompt_encoder.py:196-203."""
    c = (2 * coords01 - 1) @ G
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(G: Tensor, size: Tuple[int, int, int]) -> Tensor:
    """PositionEmbeddingRandom3D.forward prompt_encoder.py:205-219 -> [t*h*w, C] (token-major)."""
    t, h, w = size
    grid = torch.ones((t, h, w), dtype=torch.float32)
    te = (grid.cumsum(0) - 0.5) / t
    ye = (grid.cumsum(1) - 0.5) / h
    xe = (grid.cumsum(2) - 0.5) / w
    return _pe_encoding(G, torch.stack([te, xe, ye], dim=-1)).reshape(t * h * w, -1)


def _sam_attention(sd: Dict[str, Tensor], p: str, q: Tensor, k: Tensor, v: Tensor, heads: int) -> Tensor:
    """sam/transformer.py:223-245."""
    q = F.linear(q, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    k = F.linear(k, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(v, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])

    def split(x):
        b, n, c = x.shape
        return x.reshape(b, n, heads, c // heads).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    attn = torch.softmax((q @ k.permute(0, 1, 3, 2)) / math.sqrt(q.shape[-1]), dim=-1)
    out = (attn @ v).transpose(1, 2)
    out = out.reshape(out.shape[0], out.shape[1], -1)
    return F.linear(out, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def _ln(sd: Dict[str, Tensor], p: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], eps)


def two_way_transformer(sd: Dict[str, Tensor], p: str, src: Tensor, pos: Tensor, tokens: Tensor, depth: int,
                        heads: int) -> Tuple[Tensor, Tensor]:
    """TwoWayTransformer.forward sam/transformer.py:67-111 + TwoWayAttentionBlock.forward :156-187."""
    queries, keys = tokens, src
    for l in range(depth):
        lp = f"{p}layers.{l}."
        if l == 0:  # skip_first_layer_pe
            queries = _sam_attention(sd, lp + "self_attn.", queries, queries, queries, heads)
        else:
            q = queries + tokens
            queries = queries + _sam_attention(sd, lp + "self_attn.", q, q, queries, heads)
        queries = _ln(sd, lp + "norm1.", queries)
        q = queries + tokens
        k = keys + pos
        queries = _ln(sd, lp + "norm2.", queries + _sam_attention(sd, lp + "cross_attn_token_to_image.", q, k, keys, heads))
        mlp = F.linear(F.relu(F.linear(queries, sd[lp + "mlp.lin1.weight"], sd[lp + "mlp.lin1.bias"])),
                       sd[lp + "mlp.lin2.weight"], sd[lp + "mlp.lin2.bias"])
        queries = _ln(sd, lp + "norm3.", queries + mlp)
        q = queries + tokens
        k = keys + pos
        keys = _ln(sd, lp + "norm4.", keys + _sam_attention(sd, lp + "cross_attn_image_to_token.", k, q, queries, heads))
    q = queries + tokens
    k = keys + pos
    queries = _ln(sd, p + "norm_final_attn.",
                  queries + _sam_attention(sd, p + "final_attn_token_to_image.", q, k, keys, heads))
    return queries, keys


def track_single_window(sd: Dict[str, Tensor], cfg, enc_feat: Tensor, queries_n3: Tensor, labels_n: Tensor,
                        prompt_feat_nc: Optional[Tensor], prompt_label_n: Optional[Tensor],
                        task: str = "track_2d") -> Dict[str, Tensor]:
    """VideoMAETrack2DSamHead.forward / forward_single_batch (sparse_heads.py:497-667) for batch 1.

    enc_feat: [1 or N, P, C] keys (last encoder feature, plus per-query history when attending to the past).
    Returns traj [N,2,T], vis [N,1,T], depth [N,1,T], prompt_features [N,C], enc_features_history [N,P,C].
    """
    p = f"task_heads.{task}."
    N = queries_n3.shape[0]
    C = cfg.dim
    T, H, W = cfg.frames, cfg.img, cfg.img
    G = sd[p + "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    # ---- PromptEncoder._embed_points with pad=True (prompt_encoder.py:99-121) ----
    pts = torch.cat([queries_n3[:, None, :], torch.zeros(N, 1, 3)], dim=1)
    lab = torch.cat([labels_n[:, None].float(), -torch.ones(N, 1)], dim=1)
    c01 = pts.clone()
    c01[:, :, 0] = c01[:, :, 0] / T
    c01[:, :, 1] = c01[:, :, 1] / W
    c01[:, :, 2] = c01[:, :, 2] / H
    pe = _pe_encoding(G, c01.float())
    pe[lab == -1] = 0.0
    pe[lab == -1] += sd[p + "prompt_encoder.not_a_point_embed.weight"]
    for i in range(2):
        pe[lab == i] += sd[f"{p}prompt_encoder.point_embeddings.{i}.weight"]
    # ---- PromptEncoder._embed_features (prompt_encoder.py:78-97) ----
    if prompt_feat_nc is None:
        prompt_feat_nc = torch.zeros(N, C)
    if prompt_label_n is None:
        prompt_label_n = torch.zeros(N)
    feat = prompt_feat_nc[:, None, :]
    fl = prompt_label_n[:, None]
    fe = torch.zeros_like(feat)
    fe[fl == 0] = feat[fl == 0] + sd[p + "prompt_encoder.prompt_feature_embeddings.0.weight"]
    fe[fl == 1] = feat[fl == 1] + sd[p + "prompt_encoder.prompt_feature_embeddings.1.weight"]
    sparse = torch.cat([pe, fe], dim=1)  # [N, 3, C]
    # ---- MaskDecoder.predict_masks (mask_decoder.py:101-141) ----
    m = p + "mask_decoder."
    tokens = torch.cat([sd[m + "mask_tokens.weight"].unsqueeze(0).expand(N, -1, -1), sparse], dim=1)  # [N, 6, C]
    src = (enc_feat.expand(N, -1, -1) if enc_feat.shape[0] == 1 else enc_feat).contiguous()  # mask_decoder.py:116-118
    pos = dense_pe(G, cfg.grid).unsqueeze(0).expand(N, -1, -1)
    hs, keys = two_way_transformer(sd, m + "transformer.", src, pos, tokens, cfg.sam_depth, cfg.sam_heads)
    hyper = []
    for i in range(3):
        x = hs[:, i, :]
        for j in range(3):
            x = F.linear(x, sd[f"{m}output_hypernetworks_mlps.{i}.layers.{j}.weight"],
                         sd[f"{m}output_hypernetworks_mlps.{i}.layers.{j}.bias"])
            if j < 2:
                x = F.relu(x)
        hyper.append(x)
    hyper_in = torch.stack(hyper, dim=1)  # [N, 3, d1]
    nt, nh, nw = cfg.grid
    vol = keys.transpose(1, 2).reshape(N, C, nt, nh, nw)
    up = F.conv_transpose3d(vol, sd[m + "output_upscaling.0.weight"], sd[m + "output_upscaling.0.bias"], stride=2)
    u = up.mean(1, keepdim=True)  # LayerNorm3d mask_decoder.py:152-157
    s = (up - u).pow(2).mean(1, keepdim=True)
    up = (up - u) / torch.sqrt(s + 1e-6)
    up = sd[m + "output_upscaling.1.weight"][:, None, None, None] * up + sd[m + "output_upscaling.1.bias"][:, None, None, None]
    up = F.gelu(up)
    up = F.gelu(F.conv_transpose3d(up, sd[m + "output_upscaling.3.weight"], sd[m + "output_upscaling.3.bias"],
                                   stride=(1, 2, 2)))
    b_, c_, t_, h_, w_ = up.shape
    masks = (hyper_in @ up.reshape(b_, c_, t_ * h_ * w_)).reshape(b_, -1, t_, h_, w_)
    logits = F.interpolate(masks, size=(T, H, W), mode="trilinear", align_corners=False)  # sparse_heads.py:645-647
    # ---- post-processing sparse_heads.py:572-589 ----
    heat = torch.softmax(logits[:, 0].reshape(N, T, 1, H * W), dim=-1)
    gx, gy = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy")
    grid = torch.stack([gx, gy], 0).reshape(2, -1) + 0.5
    xy = torch.sum(heat * grid[None, None], dim=-1)  # [N, T, 2]
    out = {
        "traj": xy.permute(0, 2, 1),
        "vis": logits[:, 1].mean(dim=[-1, -2]).unsqueeze(1),
        "depth": torch.exp(logits[:, 2].mean(dim=[-1, -2])).unsqueeze(1),
        "prompt_features": F.linear(hs[:, 5, :], sd[p + "prompt_feature_linear_layer.weight"],
                                    sd[p + "prompt_feature_linear_layer.bias"]),
        "history": F.linear(keys, sd[p + "processed_video_features_proj.weight"],
                            sd[p + "processed_video_features_proj.bias"]),
    }
    return out


def track_windowed(sd: Dict[str, Tensor], cfg, last_feats: Sequence[Tensor], queries_bn3: Tensor, labels_bn: Tensor,
                   time_strides: Sequence[int], task: str = "track_2d", trace: Optional[list] = None) -> Dict[str, Tensor]:
    """forward_windowed_core sparse_heads.py:213-495 for estimation_directions=[1], B=1, with
    prompt_using_features / attend_to_past / modify_pointlabels_for_windowing as in configs/model.yaml.
    last_feats[w] is enc_features_bpc_2dlist[w][-1] ([1,P,C]).  ``trace`` collects the per-window
    integer/boolean state (labels, validity masks, re-seeded queries) for bit-exact checks."""
    p = f"task_heads.{task}."
    assert queries_bn3.shape[0] == 1
    N = queries_bn3.shape[1]
    ws = cfg.frames
    T = int(time_strides[-1]) + ws
    traj = torch.zeros(1, N, 2, T)
    vis = -torch.ones(1, N, 1, T) * 10.0
    dep = torch.zeros(1, N, 1, T)
    pfeat = torch.zeros(N, cfg.dim)
    plab = torch.zeros(N)
    mask_tok = sd[p + "processed_video_mask_token.weight"][0]
    hist = mask_tok[None, None, :].repeat(N, cfg.tokens, 1)
    cur_q = queries_bn3[0].clone()
    cur_lab = labels_bn[0].clone().float()
    orig_q = queries_bn3[0]
    nt, nh, nw = cfg.grid
    for wi, start in enumerate(int(s) for s in time_strides):
        q_off = cur_q.clone()
        valid_t = (torch.arange(ws).repeat(N, 1).float() + start + 0.5 - q_off[:, 0:1]) >= 0  # [N, ws]
        valid_n = valid_t.sum(-1) > 0
        q_off[:, 0] -= start
        cur_lab[~valid_n] = 0
        cur_lab[valid_n] = 1
        same = (cur_q == orig_q).sum(-1) > 0  # sparse_heads.py:330-331 (any coordinate equal)
        cur_lab[same] = 1
        cur_lab[torch.logical_and(valid_n, ~same)] = 2
        keys_in = last_feats[wi][0].unsqueeze(0) + hist  # [N,P,C]
        if trace is not None:
            trace.append({"labels": cur_lab.clone(), "valid_t": valid_t.clone(), "queries": q_off.clone(),
                          "prompt_labels": plab.clone()})
        o = track_single_window(sd, cfg, keys_in, q_off, cur_lab, pfeat, plab, task)
        vt = valid_t[:, None, :]
        vis[0, :, :, start:start + ws][vt] = o["vis"][vt]
        traj[0, :, 0:1, start:start + ws][vt] = o["traj"][:, 0:1][vt]
        traj[0, :, 1:2, start:start + ws][vt] = o["traj"][:, 1:2][vt]
        dep[0, :, :, start:start + ws][vt] = o["depth"][vt]
        if wi == len(time_strides) - 1:
            continue
        nxt = int(time_strides[wi + 1])
        pfeat[valid_n] = o["prompt_features"][valid_n]
        plab[valid_n] = 1
        # memory tokens: keep the 2nd temporal half, pad with the learned mask token (sparse_heads.py:406-448)
        h5 = o["history"].reshape(N, nt, nh, nw, cfg.dim)
        hist = torch.cat([h5[:, nt // 2:], mask_tok.expand(N, nt // 2, nh, nw, cfg.dim)], dim=1).reshape(N, cfg.tokens, cfg.dim)
        # re-seed the query at the most visible overlap frame (sparse_heads.py:455-486)
        ov_vis = vis[0, :, 0, nxt:start + ws]
        ov_traj = traj[0, :, :, nxt:start + ws]
        best = torch.argmax(ov_vis, dim=-1)  # [N]
        new_q = torch.stack([best.float() + nxt + 0.5, ov_traj[torch.arange(N), 0, best], ov_traj[torch.arange(N), 1, best]], dim=-1)
        use = new_q[:, 0] > cur_q[:, 0]
        cur_q[use] = new_q[use]
        if trace is not None:
            trace[-1]["best_vis_id"] = best.clone()
            trace[-1]["reseeded"] = use.clone()
    return {f"{task}_traj_est_bn2t": traj, f"{task}_vis_est_bn1t": vis, f"{task}_depth_est_bn1t": dep}


# --------------------------------------------------------------------------------------------------
# whole model (L4P_VideoMAE.forward l4p_videomae.py:256-330)
# --------------------------------------------------------------------------------------------------
class OracleModel:
    """Functional restatement of L4P_VideoMAE(always_use_windowed_version=True, joint_alignment=True) with the
    five heads of configs/model.yaml.  ``use_intrinsics`` mirrors task_heads['camray'].use_intrinsics."""

    def __init__(self, sd: Dict[str, Tensor], cfg, use_intrinsics: bool = True, max_queries: int = 192, seam: str = "engine"):
        self.sd, self.cfg = sd, cfg
        self.use_intrinsics = use_intrinsics
        self.max_queries = max_queries
        self.seam = seam  # how the multi-window joint alignment draws its samples (oracle/joint_oracle.py)
        self.always_use_windowed_version = True  # configs/model.yaml:19; False: a 16-frame clip takes forward_single_window
        self.depth_align_type = "affine"  # configs/model.yaml (default of VideoMAEDepthDPTHead); "linear": LinearAligner mean
        self.seam_log: list = []
        # use_intrinsics=False (the shipped default): the two third-party calls of the K estimation (stand-ins pinned against
        # the reference, tools/gen_golden_intrinsics.py), or a supplied ray-grid K per batch item instead of an estimate
        self.find_h, self.rq, self.k_override = dlt_homography, rq3, None
        self.fixed_intrinsics = True  # configs/model.yaml:45; False: every frame's own K (dense_heads.py:336-344)
        self.frame_estimate = None    # fixed_intrinsics=False: estimator in place of the two cv2 calls (see rays_to_cameras_variable_intrinsics)
        self.first_window_K: Optional[Tensor] = None
        # actpost / fusion scale factors: dense_heads.py:30-31 and :269-271
        self._actpost = lambda t: ((1, 0, 0), (1, 0, 0), (0, 0, 0), (-1, -1, -1)) if t == "camray" else ((1, 2, 2), (1, 1, 1), (0, 0, 0), (-1, -1, -1))
        self._fusion = lambda t: ((1, 1, 1), (1, 1, 1), (2, 1, 1), (2, 2, 2)) if t == "camray" else ((1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2))

    # ---- dense heads: dense_heads.py:66-74,172-182,208-217,292-352 -------------------------------
    def dense_single(self, task: str, feats: Sequence[Tensor], intrinsics_b44t: Optional[Tensor], win_id: int = 0) -> Dict[str, Tensor]:
        cfg = self.cfg
        osz = (16, 16, 16) if task == "camray" else None
        raw = dpt_forward(self.sd, task, feats, cfg, self._actpost(task), self._fusion(task), osz)
        if task == "flow_2d_backward":
            return {"flow_2d_backward_est_b2thw": raw[:, :2]}
        if task == "depth":
            return {"depth_est_b1thw": torch.exp(raw[:, :1])}
        if task == "dyn_mask":
            return {"dyn_mask_est_b1thw": raw}
        if task == "camray":
            T, H, W = cfg.frames, cfg.img, cfg.img
            Kest = None
            if self.use_intrinsics:
                E = rays_to_cameras(raw.float(), normalize_intrinsics(intrinsics_b44t, H, W).float())
            elif not self.fixed_intrinsics:
                E, Kest = rays_to_cameras_variable_intrinsics(raw.float(), (H, W), self.find_h, self.rq, self.frame_estimate)
            else:
                # VideoMAETraj3DDPTHead.forward with use_intrinsics=False, fixed_intrinsics=True (dense_heads.py:303-334):
                # K is estimated on the first window and reported for every later one, whose rotations use the INPUT K
                if win_id == 0:
                    self.first_window_K = None
                if self.first_window_K is None:
                    E, Kest = rays_to_cameras_fixed_intrinsics(raw.float(), (H, W), self.find_h, self.rq, self.k_override)
                    self.first_window_K = Kest.clone()
                else:
                    E = rays_to_cameras(raw.float(), normalize_intrinsics(intrinsics_b44t, H, W).float())
                    Kest = self.first_window_K.clone()
            pose = torch.linalg.inv(E.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
            out = {"traj3d_est_b16t": pose.reshape(pose.shape[0], 16, T)}
            if Kest is not None:
                out["traj3d_intrinsics_est_b16t"] = Kest.reshape(Kest.shape[0], 16, T)
            return out
        raise KeyError(task)

    def dense_windowed(self, task: str, feats2d: Sequence[Sequence[Tensor]], strides: Sequence[int],
                       intrinsics_b44t: Tensor) -> Dict[str, Tensor]:
        """VideoMAEFlowDPTHead.forward_windowed dense_heads.py:76-143."""
        ws = self.cfg.frames
        T = int(strides[-1]) + ws
        buf, key = None, None
        for wi, st in enumerate(int(s) for s in strides):
            o = self.dense_single(task, feats2d[wi], intrinsics_b44t[..., st:st + ws], wi)
            key = next(k for k in o if "intrinsics" not in k)
            out = o[key]
            if buf is None:
                shp = list(out.shape)
                shp[2] = T
                buf = torch.zeros(*shp)
            if wi > 0 and task == "depth":
                ov = int(strides[wi - 1]) + ws - st
                if self.depth_align_type == "linear":  # VideoMAEDepthDPTHead(align_type="linear"), dense_heads.py:158,166
                    out = linear_mean_apply(out, linear_mean_solve(out[:, :, :ov], buf[:, :, st:st + ov]))
                else:
                    sol = lstsq_affine_solve(out[:, :, :ov], buf[:, :, st:st + ov])
                    out = lstsq_affine_apply(out, sol)
            if task == "flow_2d_backward" and wi > 0:
                buf[:, :, st + 1:st + ws] = out[:, :, 1:]
            else:
                buf[:, :, st:st + ws] = out
        return {key: buf}

    def joint_depth_camray(self, feats2d, strides, intrinsics_b44t) -> Dict[str, Tensor]:
        """joint_windowed_estimation dense_heads.py:360-492.  One window: the two heads, K echoed (:419-422).  Several
        windows: oracle/joint_oracle.py — point maps, q98 threshold, similarity, apply; the two random draws of the
        reference are replaced as ``self.seam`` says ("engine": the engine's deterministic sampler / trial schedule;
        "fixed": the stand-ins the flow was pinned with against the reference, tools/gen_golden_joint.py)."""
        from oracle import joint_oracle as jo

        ws = self.cfg.frames

        def heads(win_id: int) -> Dict[str, Tensor]:
            st = int(strides[win_id])
            K = intrinsics_b44t[..., st:st + ws]
            d = self.dense_single("depth", feats2d[win_id], K)["depth_est_b1thw"]
            co = self.dense_single("camray", feats2d[win_id], K, win_id)
            c = co["traj3d_est_b16t"]
            # no estimated K when use_intrinsics: the input K is echoed (dense_heads.py:419-422)
            return {"depth": d, "camray": c,
                    "camray_intrinsics_est": co.get("traj3d_intrinsics_est_b16t", K.clone().reshape(1, 16, ws))}

        self.seam_log = []
        est = jo.joint_windowed(heads, strides, ws, self.seam, self.seam_log)
        return {"depth_est_b1thw": est["depth"], "traj3d_est_b16t": est["camray"],
                "traj3d_intrinsics_est_b16t": est["camray_intrinsics_est"]}

    def track(self, feats2d, strides, queries_bn3: Tensor, labels_bn: Tensor, trace=None) -> Dict[str, Tensor]:
        """forward_windowed sparse_heads.py:162-211 (chunks of max_queries)."""
        last = [f[-1] for f in feats2d]
        N = queries_bn3.shape[1]
        if N < self.max_queries:
            return track_windowed(self.sd, self.cfg, last, queries_bn3, labels_bn, strides, trace=trace)
        outs = []
        for i in range(int(math.ceil(N / self.max_queries))):
            sl = slice(i * self.max_queries, (i + 1) * self.max_queries)
            outs.append(track_windowed(self.sd, self.cfg, last, queries_bn3[:, sl], labels_bn[:, sl], strides, trace=trace))
        return {k: torch.cat([o[k] for o in outs], dim=1) for k in outs[0]}

    def forward_single_window(self, batch: Dict[str, Tensor], tasks: Sequence[str]) -> Dict[str, Tensor]:
        """L4P_VideoMAE.forward_single_window l4p_videomae.py:234-254 (always_use_windowed_version=False, T == 16): every
        head's plain forward on one window; the tracker's is sparse_heads.py:497-600 — raw last feature as keys (no
        history term), the caller's labels, zero prompt features, no validity masking."""
        cfg = self.cfg
        feats = encoder_forward(self.sd, batch["rgb_b3thw"], cfg)
        out: Dict[str, Tensor] = {}
        for task in tasks:
            if task == "track_2d":
                o = track_single_window(self.sd, cfg, feats[-1], batch["track_2d_pointquerries_bn3"][0],
                                        batch["track_2d_pointlabels_bn"][0], None, None)
                out.update({"track_2d_traj_est_bn2t": o["traj"][None], "track_2d_vis_est_bn1t": o["vis"][None],
                            "track_2d_depth_est_bn1t": o["depth"][None], "track_2d_prompt_features_bnc": o["prompt_features"][None],
                            # processed video tokens with the track history, projected (sparse_heads.py:560-569, :658-665)
                            "track_2d_enc_features_with_track_history_bnpc": o["history"][None]})
            else:
                out.update(self.dense_single(task, feats, batch["intrinsics_b44t"]))
        return out

    def forward(self, batch: Dict[str, Tensor], tasks: Sequence[str], trace=None) -> Dict[str, Tensor]:
        cfg = self.cfg
        rgb = batch["rgb_b3thw"]
        B, _, T, H, W = rgb.shape
        assert H == cfg.img and W == cfg.img
        if not self.always_use_windowed_version and T == cfg.frames:
            return self.forward_single_window(batch, tasks)
        assert T % 8 == 0
        strides = list(range(0, T - cfg.frames + 1, 8))
        feats2d = [encoder_forward(self.sd, rgb[:, :, s:s + cfg.frames], cfg) for s in strides]
        out: Dict[str, Tensor] = {}
        K = batch["intrinsics_b44t"]
        if "depth" in tasks and "camray" in tasks:
            for task in ("track_2d", "dyn_mask", "flow_2d_backward"):
                if task in tasks:
                    out.update(self._one(task, feats2d, strides, batch, trace))
            out.update(self.joint_depth_camray(feats2d, strides, K))
        else:
            for task in tasks:
                out.update(self._one(task, feats2d, strides, batch, trace))
        return out

    def _one(self, task, feats2d, strides, batch, trace):
        if task == "track_2d":
            return self.track(feats2d, strides, batch["track_2d_pointquerries_bn3"], batch["track_2d_pointlabels_bn"], trace)
        return self.dense_windowed(task, feats2d, strides, batch["intrinsics_b44t"])
