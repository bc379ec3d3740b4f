"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the reference's demo-side clip preparation.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (l4p_amd/data) never does.

What it restates ("host/data side of demo.py"), reference file:line per function:
  * VideoDataset.getitem_helper    l4p/data/video_dataset.py:70-135   resize-blur-resize per frame, to_tensor,
                                                                       max_frames / stride, dummy intrinsics
  * L4PDataset.__getitem__         l4p/data/l4p_dataset_mini.py:543-587 mirror-pad, resize, crop, queries, normalise
  * L4PDataset.mirror_and_pad      l4p/data/l4p_dataset_mini.py:126-190
  * L4PDataset.resize              l4p/data/l4p_dataset_mini.py:236-288
  * L4PDataset.crop                l4p/data/l4p_dataset_mini.py:290-391
  * L4PDataset.sample_tracks       l4p/data/l4p_dataset_mini.py:418-497 (grid sampling, version "uniform")

Third-party arithmetic on this path:
  * PIL.Image.resize(BILINEAR) — Pillow, unpinned in the reference's env/requirements.txt; this image has Pillow
    12.2.0.  `pil_coeffs` / `pil_resize_u8` restate the published algorithm of Pillow's libImaging/Resample.c
    (precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc: triangle filter,
    support scaled by the down-scale factor, 22-bit fixed-point coefficients, uint8 between the passes).
    Pinned bit-exactly against Pillow 12.2.0 itself by tests/test_preprocess_cpu.py (PIL is importable here and on
    the GPU box) and by the committed fixture tests/golden/preprocess_clip.npz.
  * torch.nn.functional.interpolate(mode="trilinear", align_corners=False) with an unchanged frame count =
    per-frame bilinear; restated in `interp_bilinear` (float32, ATen's source-index rule, the index a single fma),
    pinned against torch.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c: 8 bits of pixel, 2 bits of head-room for the accumulation

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)  # l4p_dataset_mini.py:103-104
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def _triangle(x: float) -> float:
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def pil_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter over the full axis.
    Returns bounds int32 [out,2] (first tap, tap count), coefficients int32 [out,ksize], ksize."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = [_triangle((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _resample_axis_u8(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One 8-bit pass (ImagingResampleHorizontal_8bpc / Vertical_8bpc): int32 accumulate, +half, >> 22, clamp."""
    bounds, kk, _ = pil_coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_u8(img_hwc: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """Image.resize((out_w, out_h), BILINEAR) of an 8-bit image: horizontal pass, then vertical, each only if that
    axis changes (ImagingResample)."""
    x = img_hwc
    if out_w != x.shape[1]:
        x = _resample_axis_u8(x, out_w, 1)
    if out_h != x.shape[0]:
        x = _resample_axis_u8(x, out_h, 0)
    return x.copy() if x is img_hwc else x


def resize_blur_resize(frame_hwc: np.ndarray, resize_size: Tuple[int, int]) -> np.ndarray:
    """video_dataset.py:86-92: PIL sizes are (width, height); the reference hands its (H, W) resize_size to PIL
    unswapped, restated as is."""
    h, w = frame_hwc.shape[:2]
    small = pil_resize_u8(frame_hwc, int(resize_size[0]), int(resize_size[1]))
    return pil_resize_u8(small, w, h)


def interp_axis(n_in: int, n_out: int):
    """ATen area_pixel_compute_source_index (align_corners=False) + guard_index_and_lambda for one axis, float32:
    returns i0, i1, w0, w1 per output position."""
    scale = np.float32(n_in) / np.float32(n_out)
    # ATen's CPU kernel evaluates scale * (dst + 0.5) - 0.5 as ONE fused multiply-add (measured: with two roundings
    # the weights are off by an ulp of the source coordinate, 8e-6 at 128); emulated through the exact f64 product
    src = (np.float64(scale) * (np.arange(n_out, dtype=np.float64) + 0.5) - 0.5).astype(np.float32)
    src = np.maximum(src, np.float32(0.0)).astype(np.float32)
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, (np.float32(1.0) - l1).astype(np.float32), l1


def interp_bilinear(x_cthw: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """F.interpolate(x[None], (T, out_h, out_w), mode="trilinear") with T unchanged (l4p_dataset_mini.py:255):
    src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out in float32, neighbour index clamped at in - 1."""
    C, T, H, W = x_cthw.shape
    y0, y1, wy0, wy1 = interp_axis(H, out_h)
    x0, x1, wx0, wx1 = interp_axis(W, out_w)
    x = x_cthw.astype(np.float32)
    top = x[:, :, y0][:, :, :, x0] * wx0 + x[:, :, y0][:, :, :, x1] * wx1
    bot = x[:, :, y1][:, :, :, x0] * wx0 + x[:, :, y1][:, :, :, x1] * wx1
    return (top * wy0[:, None] + bot * wy1[:, None]).astype(np.float32)


def mirror_pad_indices(n_frames: int, t_target: int) -> List[int]:
    """Source frame of every frame after `while T < t_target: x = cat([x, flip(x)[1:]])` (l4p_dataset_mini.py:556-559,
    175) — or a repeat of the single frame (:553-554)."""
    if n_frames == 1:
        return [0] * t_target
    idx = list(range(n_frames))
    while len(idx) < t_target:
        idx = idx + idx[::-1][1:]
    return idx


def grid_queries(spacing: float, T: int, H: int, W: int) -> np.ndarray:
    """sample_tracks, version "uniform" (l4p_dataset_mini.py:438-490): (t, x, y) = round(u * (size - 1)) + 0.5 with
    t = 0, u on the [0,1) grid of the given spacing, meshgrid indexing "xy"."""
    import torch  # torch.arange's float32 stepping is part of the behaviour

    g = torch.arange(0, 1, spacing)
    gx, gy = torch.meshgrid(g, g, indexing="xy")
    q = torch.cat([torch.zeros_like(gx)[..., None], gx[..., None], gy[..., None]], dim=-1).reshape(-1, 3).to(torch.float32)
    q[..., 0] = 0
    for i, size in enumerate((T, W, H)):
        q[..., i] = torch.round(q[..., i] * (size - 1)) + 0.5
    return q.numpy()


def preprocess_clip(frames_thwc: np.ndarray, crop_size: Optional[Tuple[int, int, int]] = (64, 224, 224),
                    resize_size: Tuple[int, int] = (224, 224), max_frames: int = 192, stride: int = 1,
                    spacing: float = 0.04, antialias: bool = True) -> Dict[str, np.ndarray]:
    """VideoDataset.__getitem__ for decoded uint8 frames [T,H,W,3] (everything after media.VideoReader)."""
    frames = frames_thwc[: max_frames - 1]  # video_dataset.py:102-103: the loop stops after max_frames - 1 frames
    if antialias:
        frames = np.stack([resize_blur_resize(f, resize_size) for f in frames])
    frames = frames[::stride]
    T0, H, W = frames.shape[:3]
    rgb = np.transpose(frames.astype(np.float32) / np.float32(255.0), (3, 0, 1, 2))  # F.to_tensor; [3,T,H,W]
    K = np.array([[min(H, W), 0, W / 2, 0], [0, min(H, W), H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)

    if crop_size is None:  # l4p_dataset_mini.py:550-552
        crop_size = (int(math.ceil(max(T0, 16) / 8) * 8), 224, 224)
    idx = mirror_pad_indices(T0, crop_size[0])
    rgb = rgb[:, idx]
    T = len(idx)
    intr = np.repeat(K[:, :, None], T, axis=2)
    # resize (:236-288)
    rh, rw = resize_size
    fy, fx = np.float32(rh / H), np.float32(rw / W)
    if not (rh / H == 1.0 and rw / W == 1.0):
        rgb = interp_bilinear(rgb, rh, rw)
        intr[0, 0] *= fx
        intr[1, 1] *= fy
        intr[0, 2] = (intr[0, 2] + np.float32(0.5)) * fx - np.float32(0.5)
        intr[1, 2] = (intr[1, 2] + np.float32(0.5)) * fy - np.float32(0.5)
    # centre crop, t0 = 0 (:290-391 with center_crop=True, start_crop_time=True)
    Tn, Hn, Wn = crop_size
    assert T >= Tn and rgb.shape[2] >= Hn and rgb.shape[3] >= Wn, "Cropping Error"
    i0, j0 = int((rgb.shape[2] - Hn) * 0.5), int((rgb.shape[3] - Wn) * 0.5)
    rgb = rgb[:, :Tn, i0:i0 + Hn, j0:j0 + Wn]
    intr = intr[:, :, :Tn].copy()
    intr[0, 2] -= j0
    intr[1, 2] -= i0
    q = grid_queries(spacing, Tn, Hn, Wn)
    rgb = (rgb - IMAGENET_MEAN[:, None, None, None]) / IMAGENET_STD[:, None, None, None]
    return dict(rgb_b3thw=np.ascontiguousarray(rgb.astype(np.float32)), intrinsics_b44t=intr,
                track_2d_pointquerries_bn3=q, track_2d_pointlabels_bn=np.ones(q.shape[0], dtype=np.float32),
                ori_video_len=np.int64(T0))
