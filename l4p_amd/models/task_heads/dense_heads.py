"""Dense DPT heads on the MI355X engine — host mirror of l4p/models/task_heads/dense_heads.py.

Same class names, constructor arguments, ``forward`` / ``forward_windowed`` signatures and output keys
as the reference; the arithmetic runs in libl4p_hip.so through l4p_amd.ops (channels-last tensors,
implicit-GEMM 3x3x3 convs, ConvTranspose-as-GEMM, fused ReLU / bias / skip epilogues).

One algebraic re-ordering versus the reference graph: FeatureFusionBlock's ``out_conv`` (a 1x1x1 conv)
is applied BEFORE the trilinear up-sampling instead of after it (dpt_block.py:229-237).  Both are
linear and the interpolation weights sum to one, so the result is identical up to float rounding,
while the conv runs on 2-8x fewer voxels.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ... import ops
from ..._lib import ACT_NONE, ACT_RELU
from ...weights import ModelCfg, actpost_of, fusion_of
from ..aligner import LstSqAffineAligner, LinearAligner, WindowOverlapAligner


class _Runtime:
    """What a head needs at run time; attached by L4P_VideoMAE once weights are loaded."""

    def __init__(self, cfg: ModelCfg, weights, dtype: int):
        self.cfg, self.weights, self.dtype = cfg, weights, dtype


def _conv_rcu(W, key: str, x: torch.Tensor, x_relu: torch.Tensor, feat: int, extra: Optional[torch.Tensor] = None,
              relu_copy: bool = False):
    """ResidualConvUnit_custom (dpt_block.py:131-157): conv2(relu(conv1(relu(x)))) + x  [+ extra].
    relu(x) comes pre-computed from the epilogue of the kernel that produced x (so conv1 streams its input by
    LDS-DMA); conv1's own ReLU is an output activation; the skip adds ride conv2's epilogue."""
    y = ops.conv3d_k3(x_relu, W[key + ".c1.w"], feat, bias=W[key + ".c1.b"], act=ACT_RELU)
    return ops.conv3d_k3(y, W[key + ".c2.w"], feat, bias=W[key + ".c2.b"], res1=x, res2=extra, relu_copy=relu_copy)


def dpt_decode(W, cfg: ModelCfg, task: str, hooks: Sequence[torch.Tensor], out_ch: int,
               output_size: Optional[Tuple[int, int, int]], image_size: Tuple[int, int, int], post_exp: bool) -> torch.Tensor:
    """DPTOutputAdapter_fix.forward (dpt_head.py:41-86) on channels-last device tensors.
    hooks: 4 x [B, P, C] engine-dtype features of layers cfg.hooks.  Returns float [B, out_ch, T, H, W]."""
    pre = f"dpt.{task}."
    Wt = lambda k: W[pre + k]
    nt, nh, nw = cfg.grid
    B = hooks[0].shape[0]
    F_ = cfg.feature_dim
    ap, fu = actpost_of(task), fusion_of(task)
    layers: List[torch.Tensor] = []
    for i in range(4):
        Li = cfg.layer_dims[i]
        a, _ = ops.gemm(hooks[i].reshape(B * cfg.tokens, cfg.dim), Wt(f"act{i}.0.w"), Li, bias=Wt(f"act{i}.0.b"))
        a = a.view(B, nt, nh, nw, Li)
        sf = ap[i]
        if any(s > 0 for s in sf):
            a = ops.conv_transpose(a, Wt(f"act{i}.1.w"), Li, tuple(2 ** s for s in sf), bias_taps=Wt(f"act{i}.1.b"))
        elif any(s < 0 for s in sf):
            a = ops.conv3d_k3(a, Wt(f"act{i}.1.w"), Li, stride=tuple(2 ** (-s) for s in sf), bias=Wt(f"act{i}.1.b"))
        layers.append(ops.conv3d_k3(a, Wt(f"rn{i}.w"), F_, relu_copy=True))  # (x, relu(x))

    def fuse(r: int, x0, x1, scale: Sequence[int]) -> torch.Tensor:
        key = f"{pre}ref{r}"
        if x1 is not None:
            out, out_relu = _conv_rcu(W, key + ".rcu1", x1[0], x1[1], F_, extra=x0, relu_copy=True)  # x0 + RCU1(x1)
        else:
            out, out_relu = x0
        out = _conv_rcu(W, key + ".rcu2", out, out_relu, F_)
        b_, t_, h_, w_, _ = out.shape
        o, _ = ops.gemm(out.view(-1, F_), W[key + ".out.w"], F_, bias=W[key + ".out.b"])
        o = o.view(b_, t_, h_, w_, F_)
        return ops.upsample_trilinear(o, (t_ * scale[0], h_ * scale[1], w_ * scale[2]), align_corners=True)

    p4 = fuse(4, layers[3], None, fu[3])
    if p4.shape[1] != layers[2][0].shape[1] or p4.shape[2] != layers[2][0].shape[2]:
        p4 = p4[:, : layers[2][0].shape[1], : layers[2][0].shape[2]].contiguous()  # dpt_head.py:70-72
    p3 = fuse(3, p4, layers[2], fu[2])
    p2 = fuse(2, p3, layers[1], fu[1])
    p1 = fuse(1, p2, layers[0], fu[0])
    h = ops.conv3d_k3(p1, Wt("head1.w"), F_ // 2, bias=Wt("head1.b"))
    osz = tuple(image_size) if output_size is None else tuple(output_size)
    h = ops.upsample_trilinear(h, osz, align_corners=True)
    h = ops.conv3d_k3(h, Wt("head2.w"), cfg.last_dim, bias=Wt("head2.b"), act=ACT_RELU)
    return ops.head_out(h, Wt("out.w"), Wt("out.b"), post_exp)


class VideoMAEFlowDPTHead(torch.nn.Module):
    """2D optical-flow DPT head (dense_heads.py:20-143)."""

    def __init__(
        self,
        task_name: str,
        out_nchan: int = 2,
        depth: int = 40,
        embed_dim: int = 1408,
        hooks_idx: Optional[List[int]] = None,
        actpost_scale_factors: Tuple[Tuple, ...] = ((1, 2, 2), (1, 1, 1), (0, 0, 0), (-1, -1, -1)),
        fusion_scale_factors: Tuple[Tuple, ...] = ((1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2)),
        output_size: Optional[Tuple[int, int, int]] = None,
        overlap_aligner_type=None,
        aligner_kwargs: Dict = {},
    ) -> None:
        super().__init__()
        self.task_name = task_name
        self.out_nchan = out_nchan
        self.depth, self.embed_dim = depth, embed_dim
        self.hooks_idx = list(hooks_idx) if hooks_idx is not None else [depth * 2 // 5, depth * 3 // 5, depth * 4 // 5, depth]
        self.actpost_scale_factors = actpost_scale_factors
        self.fusion_scale_factors = fusion_scale_factors
        self.output_size = output_size
        self.overlap_aligner_type = overlap_aligner_type
        self.aligner_kwargs = aligner_kwargs
        self.task_suffix = f"b{out_nchan}thw"
        self._rt: Optional[_Runtime] = None
        self._engine_task = ""  # key under which the ModuleDict holds this head (== packed weight prefix)
        self._post_exp = False

    # -- engine plumbing -------------------------------------------------------------------------
    def required_taps(self) -> List[int]:
        return list(self.hooks_idx)

    def _decode(self, enc_features_bpc_list, img_info) -> torch.Tensor:
        pre = getattr(enc_features_bpc_list, "decoded", None)
        if pre is not None and self._engine_task in pre:
            return pre[self._engine_task]  # window decoded earlier / on another rank (parallel.DecodedWindow)
        if self._rt is None:
            raise RuntimeError(f"head '{self.task_name}' has no weights: call load_state_dict on the model first")
        hooks = [enc_features_bpc_list.T(h) for h in self.hooks_idx]
        eng = getattr(self._rt, "engine", None)
        if eng is None or os.environ.get("L4P_DPT_PYTHON"):
            # kernel-by-kernel composition from Python (the readable statement of the graph; same kernels)
            return dpt_decode(self._rt.weights, self._rt.cfg, self._engine_task, hooks, self.out_nchan, self.output_size,
                              tuple(img_info), self._post_exp)
        out_ch = self._rt.weights[f"dpt.{self._engine_task}.out.w"].shape[0]
        osz = tuple(img_info) if self.output_size is None else tuple(self.output_size)
        return eng.dpt_forward(self._engine_task, hooks, out_ch, actpost_of(self._engine_task), fusion_of(self._engine_task),
                               osz, self._post_exp)

    # -- reference API ---------------------------------------------------------------------------
    def forward(self, enc_features_bpc_list, img_info: Tuple[int, int, int] = (16, 224, 224), **kwargs) -> Dict[str, torch.Tensor]:
        task_out = self._decode(enc_features_bpc_list, img_info)
        return {f"{self.task_name}_est_{self.task_suffix}": task_out[:, : self.out_nchan]}

    def forward_windowed(self, enc_features_bpc_2dlist, img_info: Tuple[int, int, int] = (16, 224, 224),
                         time_strides: Optional[torch.Tensor] = None, intrinsics_b44t: Optional[torch.Tensor] = None,
                         **kwargs) -> Dict[str, torch.Tensor]:
        """Per-window decode + stitching (dense_heads.py:76-143)."""
        if time_strides is None:
            return self.forward(enc_features_bpc_2dlist[0], img_info=img_info, intrinsics_b44t=intrinsics_b44t, **kwargs)
        window_size = img_info[0] if self.output_size is None else self.output_size[0]
        T = int(time_strides[-1]) + window_size
        key = f"{self.task_name}_est_{self.task_suffix}"
        est = None
        for win_id in range(len(time_strides)):
            kwargs["win_id"] = win_id
            st = int(time_strides[win_id])
            cur = self.forward(
                enc_features_bpc_2dlist[win_id], img_info=img_info,
                intrinsics_b44t=None if intrinsics_b44t is None else intrinsics_b44t[..., st:st + window_size], **kwargs)
            out = cur[key]
            if est is None:
                shp = list(out.shape)
                shp[2] = T
                est = torch.zeros(*shp, dtype=out.dtype, device=out.device)
            if win_id > 0 and self.overlap_aligner_type is not None:
                aligner = self.overlap_aligner_type(**self.aligner_kwargs)
                ov = int(time_strides[win_id - 1]) + window_size - st
                aligner.solve(out[:, :, :ov], est[:, :, st:st + ov],
                              None if intrinsics_b44t is None else intrinsics_b44t[..., st:st + ov], img_info)
                out = aligner.apply(out)
            if self.task_name == "flow_2d_backward" and win_id > 0:
                est[:, :, st + 1:st + window_size] = out[:, :, 1:]  # first frame of a later window is invalid
            else:
                est[:, :, st:st + window_size] = out
        return {key: est}


class VideoMAEDepthDPTHead(VideoMAEFlowDPTHead):
    """Depth DPT head (dense_heads.py:146-182): exp() on the decoder output, LstSq/linear seam aligner."""

    def __init__(self, task_name: str, out_nchan: int = 1, depth: int = 40, embed_dim: int = 1408,
                 depth_fn: str = "linear", hooks_idx: Optional[List[int]] = None,
                 align_window_overlap_fn: Optional[str] = None, align_type: str = "affine") -> None:
        if align_type not in ("affine", "linear"):
            raise ValueError(f"align_type={align_type!r}: expected 'affine' or 'linear'")
        super().__init__(task_name, out_nchan, depth, embed_dim, hooks_idx,
                         overlap_aligner_type=LstSqAffineAligner if align_type == "affine" else LinearAligner,
                         aligner_kwargs=dict(pre_post_fn=align_window_overlap_fn))
        if depth_fn not in ("exp", "linear"):
            raise NotImplementedError(f"depth_fn={depth_fn!r}: the engine fuses 'exp' or 'linear' into the output kernel")
        self.depth_fn = depth_fn
        self._post_exp = depth_fn == "exp"


class VideoMAEDynMaskDPTHead(VideoMAEFlowDPTHead):
    """Dynamic-mask DPT head (dense_heads.py:185-217)."""

    def __init__(self, task_name: str, out_nchan: int = 1, depth: int = 40, embed_dim: int = 1408,
                 apply_fn: str = "linear", hooks_idx: Optional[List[int]] = None) -> None:
        super().__init__(task_name, out_nchan, depth, embed_dim, hooks_idx, overlap_aligner_type=None)
        if apply_fn != "linear":
            raise NotImplementedError(f"apply_fn={apply_fn!r} is not used by configs/model.yaml")
        self.apply_fn = apply_fn

    def forward(self, enc_features_bpc_list, img_info=(16, 224, 224), **kwargs):
        return {f"{self.task_name}_est_{self.task_suffix}": self._decode(enc_features_bpc_list, img_info)}


class VideoMAETraj3DDPTHead(VideoMAEFlowDPTHead):
    """Ray-map head -> camera poses (dense_heads.py:257-352)."""

    def __init__(self, task_name: str, depth: int = 40, embed_dim: int = 1408, hooks_idx: Optional[List[int]] = None,
                 actpost_scale_factors=((1, 0, 0), (1, 0, 0), (0, 0, 0), (-1, -1, -1)),
                 fusion_scale_factors=((1, 1, 1), (1, 1, 1), (2, 1, 1), (2, 2, 2)),
                 output_size: Optional[Tuple[int, int, int]] = (16, 16, 16), use_intrinsics: bool = True,
                 fixed_intrinsics: bool = False) -> None:
        super().__init__(task_name, 6, depth, embed_dim, hooks_idx, actpost_scale_factors, fusion_scale_factors,
                         output_size)
        self.task_suffix = "b16t"
        self.use_intrinsics = use_intrinsics
        self.fixed_intrinsics = fixed_intrinsics
        self.first_window_intrinsics_b44t = None

    def forward(self, enc_features_bpc_list, img_info=(16, 224, 224), intrinsics_b44t: Optional[torch.Tensor] = None,
                **kwargs) -> Dict[str, torch.Tensor]:
        from ...utils.geometry_utils import intrinsics_from_rays, poses_from_rays

        T, H, W = img_info
        rays = self._decode(enc_features_bpc_list, img_info)  # float [B,6,16,16,16]
        key = f"{self.task_name}_est_{self.task_suffix}"
        if self.use_intrinsics:
            if intrinsics_b44t is None:
                raise ValueError("intrinsics_b44t is required when use_intrinsics=True")
            return {key: poses_from_rays(rays, intrinsics_b44t.to(rays.device, torch.float32), H, W)}
        if not self.fixed_intrinsics:
            # per-frame intrinsics and extrinsics (dense_heads.py:336-344): every frame's own K, rotation from its RQ step
            from ...utils.geometry_utils import cameras_from_rays_variable_intrinsics

            pose, K_est = cameras_from_rays_variable_intrinsics(rays, H, W, reproj_threshold=0.2)
            return {key: pose, f"{self.task_name}_intrinsics_est_{self.task_suffix}": K_est.reshape(K_est.shape[0], 16, T)}
        # fixed intrinsics, estimated once on the first window (dense_heads.py:303-334)
        assert "win_id" in kwargs, "win_id is required when setting fixed intrinsics as True"
        if kwargs["win_id"] == 0:
            self.first_window_intrinsics_b44t = None
        if self.first_window_intrinsics_b44t is None:
            K_est = intrinsics_from_rays(rays, H, W, reproj_threshold=0.2)  # [B,4,4,T], first frame, pixel units
            pose = poses_from_rays(rays, K_est, H, W)
            self.first_window_intrinsics_b44t = K_est.clone()
        else:
            # as the reference: later windows rotate with the INPUT intrinsics (dense_heads.py:327-333) and report the
            # first window's estimate
            if intrinsics_b44t is None:
                raise ValueError("intrinsics_b44t is required for windows after the first one")
            pose = poses_from_rays(rays, intrinsics_b44t.to(rays.device, torch.float32), H, W)
            K_est = self.first_window_intrinsics_b44t.clone()
        return {key: pose, f"{self.task_name}_intrinsics_est_{self.task_suffix}": K_est.reshape(K_est.shape[0], 16, T)}


def joint_windowed_estimation(task_names: List[str], task_heads: torch.nn.ModuleDict, enc_features_bpc_2dlist,
                              time_strides: Optional[torch.Tensor] = None, intrinsics_b44t: Optional[torch.Tensor] = None,
                              img_info: Tuple[int, int, int] = (16, 224, 224), **kwargs) -> Dict[str, torch.Tensor]:
    """Joint depth + camera estimation over sliding windows (dense_heads.py:360-492): per-window heads, then a
    similarity alignment of the overlap point maps (KabaschUmeyama3DAligner) applied to pose and depth."""
    from ..aligner import KabaschUmeyama3DAligner

    out_all: Dict[str, torch.Tensor] = {}
    if time_strides is None:
        for name in task_names:
            out_all.update(task_heads[name].forward(enc_features_bpc_2dlist[0], img_info=img_info,
                                                    intrinsics_b44t=intrinsics_b44t, **kwargs))
        return out_all
    ws = img_info[0]
    T = int(time_strides[-1]) + ws
    est: Dict[str, Optional[torch.Tensor]] = {name: None for name in task_names}
    est["camray_intrinsics_est"] = None
    cam = task_heads["camray"]
    for win_id in range(len(time_strides)):
        st = int(time_strides[win_id])
        kwargs["win_id"] = win_id
        cur: Dict[str, torch.Tensor] = {}
        for name in task_names:
            head = task_heads[name]
            o = head.forward(enc_features_bpc_2dlist[win_id], img_info=img_info,
                             intrinsics_b44t=intrinsics_b44t[..., st:st + ws], **kwargs)
            cur[name] = o[f"{head.task_name}_est_{head.task_suffix}"]
            if name == "camray":
                kkey = f"{head.task_name}_intrinsics_est_{head.task_suffix}"
                # the reference hard-codes batch 1 here (dense_heads.py:421); clips are independent, so B is kept
                cur["camray_intrinsics_est"] = (o[kkey] if kkey in o else
                                                intrinsics_b44t[..., st:st + ws].clone().reshape(-1, 16, ws))
        for k, v in cur.items():
            if est[k] is None:
                shp = list(v.shape)
                shp[2] = T
                est[k] = torch.zeros(*shp, dtype=v.dtype, device=v.device)
        if win_id > 0:
            aligner = KabaschUmeyama3DAligner()
            ov = int(time_strides[win_id - 1]) + ws - st
            pred = {n: cur[n][:, :, :ov] for n in task_names}
            target = {n: est[n][:, :, st:st + ov] for n in task_names}
            pred["camray_intrinsics"] = cur["camray_intrinsics_est"][:, :, :ov].reshape(-1, 4, 4, ov).clone()
            target["camray_intrinsics"] = est["camray_intrinsics_est"][:, :, st:st + ov].reshape(-1, 4, 4, ov)
            aligner.solve(pred, target, img_info)
            cur = aligner.apply(cur)
        for k, v in cur.items():
            est[k][:, :, st:st + ws] = v
    for name in task_names:
        head = task_heads[name]
        out_all[f"{head.task_name}_est_{head.task_suffix}"] = est[name]
    out_all[f"{cam.task_name}_intrinsics_est_{cam.task_suffix}"] = est["camray_intrinsics_est"]
    return out_all
