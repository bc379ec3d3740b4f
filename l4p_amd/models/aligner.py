"""Window-overlap aligners on the GPU — host mirror of l4p/models/aligner.py.

LstSqAffineAligner / LinearAligner keep the reference's solve()/apply() protocol; the sums and the
2x2 solve run in libl4p_hip.so (geom.hip) instead of torch.linalg.lstsq on a 401k x 2 matrix.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import torch

from .. import _lib
from ..ops import _p, _stream


class WindowOverlapAligner(ABC):
    @abstractmethod
    def solve(self, pred, target, intrinsics, img_info):
        ...

    @abstractmethod
    def apply(self, pred):
        ...


def _mode(pre_post_fn: Optional[str]) -> int:
    if pre_post_fn in (None, "identity"):
        return 0
    if pre_post_fn == "inverse":
        return 1
    raise ValueError(f"Unknown pre_post_fn: {pre_post_fn}")


AFFINE_SCRATCH_DOUBLES = 4096  # L4P_AFFINE_SCRATCH_DOUBLES (include/l4p_hip.h)
ALIGN_INVERSE, ALIGN_RATIO_MEAN = 1, 2
QUANTILE_WS_UINTS = 2052  # L4P_QUANTILE_WS_UINTS


class LstSqAffineAligner(WindowOverlapAligner):
    """Scale + shift least squares between overlapping windows (aligner.py:29-66); one solve per batch item."""

    _mode_bits = 0

    def __init__(self, pre_post_fn: Optional[str] = "identity") -> None:
        self.inverse = _mode(pre_post_fn)
        self.sol: Optional[torch.Tensor] = None  # [B, 2] = (scale, shift), float, on device

    def solve(self, pred, target, intrinsics=None, img_info=None, pred_conf=None, target_conf=None):
        assert pred.is_cuda and pred.dtype == torch.float32 and pred.shape == target.shape
        lib = _lib.load()
        bs = pred.shape[0]
        self.sol = torch.empty(bs, 2, dtype=torch.float32, device=pred.device)
        scratch = torch.empty(AFFINE_SCRATCH_DOUBLES, dtype=torch.float64, device=pred.device)
        for b in range(bs):
            pb, tb = pred[b].contiguous(), target[b].contiguous()
            _lib.check(lib.l4p_affine_align_solve(_stream(), _p(pb), _p(tb), pb.numel(), self.inverse | self._mode_bits,
                                                  _p(scratch), self.sol[b].data_ptr()), "l4p_affine_align_solve")

    def apply(self, pred):
        assert self.sol is not None, "solve() first"
        lib = _lib.load()
        out = torch.empty_like(pred, memory_format=torch.contiguous_format)
        for b in range(pred.shape[0]):
            xb = pred[b].contiguous()
            _lib.check(lib.l4p_affine_align_apply(_stream(), _p(xb), out[b].data_ptr(), xb.numel(), self.inverse,
                                                  self.sol[b].data_ptr()), "l4p_affine_align_apply")
        return out


class LinearAligner(LstSqAffineAligner):
    """Scale-only aligner (aligner.py:69-118): scale = mean or median of f(target) / (f(pred) + 1e-8), shift = 0.
    ``mean``: the affine aligner's two kernels in their ratio-mean mode.  ``median``: torch.median's LOWER median of the
    ratios by exact radix selection (l4p_ratio_median_solve; csrc/umeyama.hip qsel_*)."""

    _mode_bits = ALIGN_RATIO_MEAN

    def __init__(self, pre_post_fn: Optional[str] = "identity", method: str = "mean") -> None:
        super().__init__(pre_post_fn)
        if method not in ["mean", "median"]:
            raise ValueError(f"Unknown method: {method}")
        self.method = method

    def solve(self, pred, target, intrinsics=None, img_info=None, pred_conf=None, target_conf=None):
        if self.method == "mean":
            return super().solve(pred, target, intrinsics, img_info)
        assert pred.is_cuda and pred.dtype == torch.float32 and pred.shape == target.shape
        lib = _lib.load()
        bs = pred.shape[0]
        self.sol = torch.empty(bs, 2, dtype=torch.float32, device=pred.device)
        n = pred[0].numel()
        ratios = torch.empty(n, dtype=torch.float32, device=pred.device)
        ws = torch.empty(QUANTILE_WS_UINTS, dtype=torch.int32, device=pred.device)
        for b in range(bs):
            pb, tb = pred[b].contiguous(), target[b].contiguous()
            _lib.check(lib.l4p_ratio_median_solve(_stream(), _p(pb), _p(tb), n, self.inverse, _p(ratios), _p(ws),
                                                  self.sol[b].data_ptr()), "l4p_ratio_median_solve")


class KabaschUmeyama3DAligner(WindowOverlapAligner):
    """Joint depth+pose seam aligner (aligner.py:158-265).  The reference runs skimage RANSAC on the CPU
    (unpinned, randomised); the GPU version is built in l4p_amd.utils.umeyama."""

    def __init__(self, calc_scale: bool = True) -> None:
        self.rel_T_b44 = None
        self.calc_scale = calc_scale
        self.min_samples = 10
        self.reprojection_threshold = 0.01
        self.confidence = 0.99
        self.frame_sample_step = 3
        self.point_sample_ratio = 0.1

    def solve(self, pred, target, img_info):
        from ..utils.umeyama import solve_window_similarity

        self.rel_T_b44 = solve_window_similarity(self, pred, target, img_info)

    def apply(self, pred):
        assert self.rel_T_b44 is not None, "rel_T_b44 is not set"
        from ..utils.umeyama import apply_window_similarity

        return apply_window_similarity(self.rel_T_b44, pred)
