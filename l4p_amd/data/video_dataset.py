"""`VideoDataset` — the reference's generic-video dataset (l4p/data/video_dataset.py:18-137 on top of
L4PDataset, l4p/data/l4p_dataset_mini.py:48-587) with the per-frame work moved to the GPU.

Same constructor arguments, same sample dict (keys, shapes, dtypes, value conventions) as the reference; tensors
live on the GPU.  What runs where:
  * video decoding stays outside (mediapy / ffmpeg when installed; or pass decoded uint8 frames with ``frames=``);
  * the per-pixel work — Pillow's resize-blur-resize (:86-92), to_tensor, temporal mirror-padding, the trilinear
    resize, the centre crop and the ImageNet normalisation — is four launches of libl4p_hip.so
    (csrc/preprocess.hip: three integer resample passes + one fused resize / crop / normalise kernel);
  * the tiny control values (frame index table, dummy intrinsics and their rescaling, grid queries, dummy ground
    truth) are computed on the host with the reference's own formulas.

The product path never imports oracle/ and raises L4PHipError when the HIP library is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from math import ceil
from typing import Dict, List, Literal, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import _lib
from ..ops import _p, _stream

ESTIMATION_DIRECTIONS = Literal[1, -1]
_MEAN = (0.485, 0.456, 0.406)  # l4p_dataset_mini.py:103-104
_STD = (0.229, 0.224, 0.225)

_tables: Dict[Tuple[int, int, str], Tuple[torch.Tensor, torch.Tensor, int]] = {}


def _pil_tables(in_size: int, out_size: int, device: torch.device) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """Pillow's BILINEAR coefficient tables for one axis on the device (cached): l4p_pil_coeffs (host) + one copy."""
    key = (in_size, out_size, str(device))
    hit = _tables.get(key)
    if hit is not None:
        return hit
    lib = _lib.load()
    ks = C.c_int(0)
    _lib.check(lib.l4p_pil_coeffs(in_size, out_size, None, None, 0, C.byref(ks)), "l4p_pil_coeffs")
    bounds = np.empty((out_size, 2), dtype=np.int32)
    kk = np.empty((out_size, ks.value), dtype=np.int32)
    _lib.check(lib.l4p_pil_coeffs(in_size, out_size, bounds.ctypes.data, kk.ctypes.data, kk.size, C.byref(ks)),
               "l4p_pil_coeffs")
    out = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ks.value)
    _tables[key] = out
    return out


def _resample(x: torch.Tensor, axis: int, out_size: int, cols: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One Pillow pass over uint8 images [n,h,w,c]; axis 1 = width, 0 = height.  ``cols`` (device int64 indices): produce
    only these output positions of the axis, in this order (the pass is a table-driven gather: rows of the tables)."""
    n, h, w, c = x.shape
    bounds, kk, ks = _pil_tables(w if axis == 1 else h, out_size, x.device)
    if cols is not None:  # gathered rows of the tables, cached with the column list (an attribute set by _x_columns)
        key = (w if axis == 1 else h, out_size, str(x.device), getattr(cols, "_l4p_key", None))
        hit = _tables.get(key) if key[3] is not None else None
        if hit is None:
            hit = (bounds[cols].contiguous(), kk[cols].contiguous(), ks)
            if key[3] is not None:
                _tables[key] = hit
        bounds, kk, out_size = hit[0], hit[1], int(cols.numel())
    out = torch.empty((n, h, out_size, c) if axis == 1 else (n, out_size, w, c), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().l4p_pil_resample_u8(_stream(), _p(x), _p(out), n, h, w, c, axis, out_size, _p(bounds), _p(kk), ks),
               "l4p_pil_resample_u8")
    return out


_xcols: Dict[Tuple[int, int, int, int, str], Tuple[torch.Tensor, torch.Tensor]] = {}


def _x_columns(W: int, res_w: int, j0: int, Wn: int, device: torch.device) -> Tuple[torch.Tensor, torch.Tensor]:
    """The two source columns (interleaved, int64 [2*Wn]) and the lerp weight (float [Wn]) of every output column of the
    resize W -> res_w cropped at j0, from the library's own index rule (l4p_resize_index_table); cached."""
    key = (W, res_w, j0, Wn, str(device))
    hit = _xcols.get(key)
    if hit is None:
        xi0, xi1 = np.empty(Wn, dtype=np.int32), np.empty(Wn, dtype=np.int32)
        xlam = np.empty(Wn, dtype=np.float32)
        _lib.check(_lib.load().l4p_resize_index_table(W, res_w, j0, Wn, xi0.ctypes.data, xi1.ctypes.data, xlam.ctypes.data),
                   "l4p_resize_index_table")
        cols = torch.from_numpy(np.stack([xi0, xi1], axis=1).reshape(-1).astype(np.int64)).to(device)
        cols._l4p_key = key
        hit = (cols, torch.from_numpy(xlam).to(device))
        _xcols[key] = hit
    return hit


def pil_resize_blur_resize(frames: torch.Tensor, pil_size: Tuple[int, int], keep_last_pass: bool = True,
                           x_cols: Optional[torch.Tensor] = None):
    """video_dataset.py:86-92 for uint8 frames [n,H,W,3] on the GPU: Image.resize(pil_size = (width, height), BILINEAR)
    and back to (W, H).  With ``keep_last_pass=False`` the final vertical pass is NOT run: returns (rows, tables) with
    rows [n,ph,W,3] and the (bounds, coeffs, ksize) of the pending ph -> H pass, for the fused kernel; ``x_cols`` then
    restricts the horizontal up-scaling pass to the listed columns of the full-width frame (rows [n,ph,len(x_cols),3])."""
    assert x_cols is None or not keep_last_pass
    n, H, W, _ = frames.shape
    pw, ph = int(pil_size[0]), int(pil_size[1])
    x = frames
    if pw != W:
        x = _resample(x, 1, pw)
    if ph != H:
        x = _resample(x, 0, ph)
    if pw != W:
        x = _resample(x, 1, W, cols=x_cols)
    if ph == H:
        return x if keep_last_pass else (x, None)
    if keep_last_pass:
        return _resample(x, 0, H)
    return x, _pil_tables(ph, H, frames.device)


def mirror_pad_indices(n_frames: int, t_target: int) -> List[int]:
    """Source frame of every frame after the reference's padding loop (l4p_dataset_mini.py:553-559): a single frame is
    repeated, otherwise `x = cat([x, flip(x)[1:]])` while the clip is shorter than the crop."""
    if n_frames == 1:
        return [0] * t_target
    idx = list(range(n_frames))
    while len(idx) < t_target:
        idx = idx + idx[::-1][1:]
    return idx


def grid_queries(spacing: float, T: int, H: int, W: int) -> torch.Tensor:
    """Grid queries of sample_tracks, version "uniform" (l4p_dataset_mini.py:438-490)."""
    g = torch.arange(0, 1, spacing)
    gx, gy = torch.meshgrid(g, g, indexing="xy")
    q = torch.cat([torch.zeros_like(gx)[..., None], gx[..., None], gy[..., None]], dim=-1).reshape(-1, 3).to(torch.float32)
    q[..., 0] = 0  # queries sit in the first frame
    for i, size in enumerate((T, W, H)):
        q[..., i] = torch.round(q[..., i] * (size - 1)) + 0.5
    return q


def prepare_clip(frames: torch.Tensor, crop_size: Optional[Tuple[int, int, int]], resize_size: Optional[Tuple[int, int]],
                 max_frames: int = 192, stride: int = 1, spacing: float = 0.02, seq_name: str = "",
                 default_sample_size: Tuple[int, int, int] = (16, 224, 224), length_multiply_of: int = 8) -> Dict[str, object]:
    """Decoded uint8 frames [T,H,W,3] (device) -> the sample dict of VideoDataset.__getitem__ (un-batched)."""
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise ValueError("frames must be uint8 [T,H,W,3]")
    if not frames.is_cuda:
        raise _lib.L4PHipError("prepare_clip needs the frames on the GPU (there is no CPU fallback)")
    dev = frames.device
    frames = frames[: max_frames - 1]  # video_dataset.py:102-103: the read loop stops after max_frames - 1 frames
    if stride > 1:
        frames = frames[::stride]  # the blur is per frame: skipped frames need not be blurred
    frames = frames.contiguous()
    T0, H, W, _ = frames.shape
    ori_video_len = T0

    # -- temporal bookkeeping (l4p_dataset_mini.py:546-559, crop :309-311 with start_crop_time) --
    if crop_size is None:
        crop_size = (ceil(max(T0, default_sample_size[0]) / length_multiply_of) * length_multiply_of,) + tuple(default_sample_size[1:])
    Tn, Hn, Wn = (int(v) for v in crop_size)
    idx = mirror_pad_indices(T0, Tn)
    T_pad = len(idx)
    assert T_pad >= Tn, f"Cropping Error: diff_shape {[T_pad - Tn]}"
    idx = idx[:Tn]

    # -- dummy intrinsics (video_dataset.py:113-127) and their resize / crop updates (l4p_dataset_mini.py:281-285,379-381) --
    intr = torch.Tensor([[min(H, W), 0, W / 2, 0], [0, min(H, W), H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    intr = intr[:, :, None].repeat(1, 1, Tn).clone()
    if resize_size is not None:
        res_h, res_w = int(resize_size[0]), int(resize_size[1])
    else:
        res_h, res_w = H, W
    factor = (res_h / H, res_w / W)
    if not (factor[0] == 1.0 and factor[1] == 1.0):
        intr[0, 0, :] = intr[0, 0, :] * factor[1]
        intr[1, 1, :] = intr[1, 1, :] * factor[0]
        intr[0, 2, :] = (intr[0, 2, :] + 0.5) * factor[1] - 0.5
        intr[1, 2, :] = (intr[1, 2, :] + 0.5) * factor[0] - 0.5
    diff = (res_h - Hn, res_w - Wn)
    assert diff[0] >= 0 and diff[1] >= 0, f"Cropping Error: diff_shape {list(diff)}"
    i0, j0 = int(diff[0] * 0.5), int(diff[1] * 0.5)  # center_crop
    intr[0, 2, :] = intr[0, 2, :] - j0
    intr[1, 2, :] = intr[1, 2, :] - i0

    # -- pixels: blur passes + fused resize / crop / normalise --
    pil_size = resize_size if resize_size is not None else (W, H)  # (the reference hands its (H, W) pair to PIL as is)
    used = sorted(set(idx))
    remap = {f: k for k, f in enumerate(used)}
    if len(used) < T0:
        frames = frames[torch.tensor(used, device=dev)]
    # an output column only reads two columns of the blurred frame: the horizontal up-scaling pass produces just those
    xl = None
    x_cols = None
    if int(pil_size[0]) != W and not (factor[0] == 1.0 and factor[1] == 1.0):
        x_cols, xl = _x_columns(W, res_w, j0, Wn, dev)
    rows, vt = pil_resize_blur_resize(frames, pil_size, keep_last_pass=False, x_cols=x_cols)
    fidx = torch.tensor([remap[f] for f in idx], dtype=torch.int32, device=dev)
    rgb = torch.empty((3, Tn, Hn, Wn), dtype=torch.float32, device=dev)
    mean = (C.c_float * 3)(*_MEAN)
    std = (C.c_float * 3)(*_STD)
    vb, vk, vks = (vt if vt is not None else (None, None, 0))
    _lib.check(_lib.load().l4p_clip_resize_normalize(_stream(), _p(rows), _p(fidx), _p(rgb), Tn, H, W, res_h, res_w, i0, j0, Hn,
                                                     Wn, mean, std, rows.shape[1], _p(vb), _p(vk), vks, _p(xl)),
               "l4p_clip_resize_normalize")

    # -- queries and the dummy ground truth of sample_tracks (:438-495) --
    q = grid_queries(spacing, Tn, Hn, Wn)
    N = q.shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    return {
        "rgb_b3thw": rgb,
        "intrinsics_b44t": intr.to(dev),
        "instanceseg_b1thw": torch.zeros((1, Tn, Hn, Wn), **f32),
        "track_2d_traj_bn2t": torch.zeros((N, 2, Tn), **f32),
        "track_2d_vis_bn1t": torch.zeros((N, 1, Tn), dtype=torch.bool, device=dev),
        "track_2d_depth_bn1t": torch.ones((N, 1, Tn), **f32),
        "track_2d_valid_bn1t": torch.zeros((N, 1, Tn), dtype=torch.bool, device=dev),
        "track_2d_pointquerries_bn3": q.to(dev),
        "track_2d_pointlabels_bn": torch.ones((N,), **f32),
        "rgb_mean_b3111": torch.tensor(_MEAN, **f32)[:, None, None, None],
        "rgb_std_b3111": torch.tensor(_STD, **f32)[:, None, None, None],
        "seq_name": seq_name,
        "ori_video_len": ori_video_len,
    }


class VideoDataset(torch.utils.data.Dataset):
    """Drop-in for l4p.data.video_dataset.VideoDataset (same arguments; `frames` / `device` are additions)."""

    default_sample_size = (16, 224, 224)

    def __init__(
        self,
        video_paths: List[str],
        dataset_type: str = "video",
        max_frames: int = 192,
        stride: int = 1,
        crop_size: Union[None, Tuple[int, int, int]] = None,
        resize_size: Tuple[int, int] = (224, 224),
        center_crop: bool = True,
        start_crop_time: bool = True,
        estimation_directions: Sequence[ESTIMATION_DIRECTIONS] = (1,),
        resize_mode: Dict[str, str] = {"rgb_b3thw": "trilinear"},
        track_2d_querry_sampling_spacing: float = 0.02,
        frames: Optional[Dict[str, Union[np.ndarray, torch.Tensor]]] = None,
        device: Union[str, torch.device] = "cuda",
    ):
        super().__init__()
        if not (center_crop and start_crop_time):
            raise NotImplementedError("random spatial / temporal crops are a training feature; the demo path uses centre crops")
        if resize_mode.get("rgb_b3thw", "trilinear") != "trilinear":
            raise NotImplementedError("the engine implements the reference's default resize mode (trilinear) for rgb")
        if isinstance(resize_size, int):
            resize_size = (resize_size, resize_size)
        self.video_paths = list(video_paths)
        self.dataset_type = dataset_type
        self.max_frames = max_frames
        self.stride = stride
        self.crop_size = crop_size
        self.resize_size = resize_size
        self.estimation_directions = list(estimation_directions)
        self.track_2d_querry_sampling_spacing = track_2d_querry_sampling_spacing
        self.length_multiply_of = 8
        self.frames = frames
        self.device = torch.device(device)
        self.len = len(self.video_paths)

    def __len__(self) -> int:
        return self.len

    def _decode(self, path: str) -> torch.Tensor:
        """uint8 [T,H,W,3] frames of one video (at most max_frames - 1, as the reference's read loop keeps)."""
        if self.frames is not None and path in self.frames:
            f = self.frames[path]
            f = torch.from_numpy(np.ascontiguousarray(f)) if isinstance(f, np.ndarray) else f
            return f[: self.max_frames - 1]
        try:
            import mediapy as media  # not in this image; the reference's own reader when present
        except ImportError as e:
            raise ImportError("video decoding needs mediapy (absent here): pass decoded uint8 frames with frames={path: array}") from e
        out = []
        with media.VideoReader(path) as reader:
            for rgb in reader:
                out.append(np.asarray(rgb)[..., :3])
                if len(out) == self.max_frames - 1:
                    break
        return torch.from_numpy(np.stack(out))

    def __getitem__(self, index: int) -> Dict[str, object]:
        path = self.video_paths[index]
        frames = self._decode(path).to(self.device, non_blocking=True)
        return prepare_clip(frames, self.crop_size, self.resize_size, self.max_frames, self.stride,
                            self.track_2d_querry_sampling_spacing, seq_name=str(os.path.basename(path)),
                            default_sample_size=self.default_sample_size, length_multiply_of=self.length_multiply_of)
