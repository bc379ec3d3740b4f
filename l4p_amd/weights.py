"""State-dict schema of the L4P model and deterministic name-seeded weights.

The reference ships no checkpoint in-tree (weights/download.sh pulls it from Google Drive) and this
environment has no network, so every parity run uses *name-seeded* random weights: each tensor is
drawn from ``torch.Generator().manual_seed(crc32(name))``.  The same generator feeds the reference
(tools/gen_golden.py), the oracle and the HIP engine, so nothing of the reference needs to travel.

``state_dict_schema`` restates the key set / shapes the reference produces
(reference l4p/models/l4p_videomae.py:163-185, dense_heads.py:20-64,
dpt_block.py:29-90,160-238,340-509, sparse_heads.py:104-136, sam/*.py) and is checked against the
manifest extracted from the real reference modules (tests/golden/manifest_*.json).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch


@dataclass
class ModelCfg:
    """Geometry of the shared encoder + heads.  ``full()`` is the shipped VideoMAE-v2-giant setup
    (l4p_videomae.py:163-185, configs/model.yaml); ``mini()`` is a 704-wide / 4-deep version used for
    full-tensor parity tests (all kernel alignment rules still hold)."""

    dim: int = 1408
    depth: int = 40
    heads: int = 16
    mlp_hidden: int = 6144  # int(dim * 48 / 11)
    hooks: Tuple[int, int, int, int] = (14, 21, 28, 36)
    frames: int = 16
    img: int = 224
    patch: Tuple[int, int, int] = (2, 14, 14)
    in_chans: int = 3
    ln_eps: float = 1e-6
    # DPT decoder (dense_heads.py:38-45)
    layer_dims: Tuple[int, int, int, int] = (256, 512, 1024, 1024)
    feature_dim: int = 256
    last_dim: int = 128
    # tracker (sparse_heads.py:20-38, configs/model.yaml:53-66)
    sam_depth: int = 2
    sam_heads: int = 8
    sam_mlp: int = 2048
    decoding_out_dim_factor: int = 8
    dense_tasks: Dict[str, int] = field(
        default_factory=lambda: OrderedDict(flow_2d_backward=2, depth=1, dyn_mask=1, camray=6)
    )

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads

    @property
    def tokens(self) -> int:
        return (self.frames // self.patch[0]) * (self.img // self.patch[1]) * (self.img // self.patch[2])

    @property
    def grid(self) -> Tuple[int, int, int]:
        return (self.frames // self.patch[0], self.img // self.patch[1], self.img // self.patch[2])

    def describe(self) -> str:
        """One-line identity of the geometry (stamped into packed arena files, tools/ckpt_to_arena.py)."""
        return (f"dim{self.dim}-depth{self.depth}-heads{self.heads}-mlp{self.mlp_hidden}-hooks{'.'.join(map(str, self.hooks))}-"
                f"f{self.frames}-img{self.img}-p{'x'.join(map(str, self.patch))}-sam{self.sam_depth}.{self.sam_heads}.{self.sam_mlp}")

    @staticmethod
    def full() -> "ModelCfg":
        return ModelCfg()

    @staticmethod
    def mini() -> "ModelCfg":
        return ModelCfg(dim=704, depth=4, heads=8, mlp_hidden=int(704 * 48 / 11), hooks=(1, 2, 3, 4))


# actpost / fusion scale factors: dense_heads.py:30-31 (flow/depth/dyn_mask) and :269-271 (camray)
DENSE_ACTPOST = ((1, 2, 2), (1, 1, 1), (0, 0, 0), (-1, -1, -1))
DENSE_FUSION = ((1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2))
CAMRAY_ACTPOST = ((1, 0, 0), (1, 0, 0), (0, 0, 0), (-1, -1, -1))
CAMRAY_FUSION = ((1, 1, 1), (1, 1, 1), (2, 1, 1), (2, 2, 2))


def actpost_of(task: str):
    return CAMRAY_ACTPOST if task == "camray" else DENSE_ACTPOST


def fusion_of(task: str):
    return CAMRAY_FUSION if task == "camray" else DENSE_FUSION


def encoder_schema(c: ModelCfg, prefix: str = "video_encoder.") -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    D, Hd = c.dim, c.mlp_hidden
    s[prefix + "patch_embed.proj.weight"] = (D, c.in_chans) + tuple(c.patch)
    s[prefix + "patch_embed.proj.bias"] = (D,)
    for i in range(c.depth):
        b = f"{prefix}blocks.{i}."
        s[b + "norm1.weight"] = (D,)
        s[b + "norm1.bias"] = (D,)
        s[b + "attn.q_bias"] = (D,)
        s[b + "attn.v_bias"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D)
        s[b + "attn.proj.weight"] = (D, D)
        s[b + "attn.proj.bias"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (Hd, D)
        s[b + "mlp.fc1.bias"] = (Hd,)
        s[b + "mlp.fc2.weight"] = (D, Hd)
        s[b + "mlp.fc2.bias"] = (D,)
    s[prefix + "norm.weight"] = (D,)
    s[prefix + "norm.bias"] = (D,)
    return s


def dpt_schema(c: ModelCfg, task: str, out_ch: int) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    p = f"task_heads.{task}.task_head.dpt."
    F_, L = c.feature_dim, c.layer_dims
    for i in range(4):
        s[f"{p}scratch.layer{i + 1}_rn.weight"] = (F_, L[i], 3, 3, 3)
    for i in range(4):
        s[f"{p}scratch.layer_rn.{i}.weight"] = (F_, L[i], 3, 3, 3)
    for r in (1, 2, 3, 4):
        rp = f"{p}scratch.refinenet{r}."
        s[rp + "out_conv.weight"] = (F_, F_, 1, 1, 1)
        s[rp + "out_conv.bias"] = (F_,)
        for u in (1, 2):
            for cv in (1, 2):
                s[f"{rp}resConfUnit{u}.conv{cv}.weight"] = (F_, F_, 3, 3, 3)
                s[f"{rp}resConfUnit{u}.conv{cv}.bias"] = (F_,)
    s[p + "head1.0.weight"] = (F_ // 2, F_, 3, 3, 3)
    s[p + "head1.0.bias"] = (F_ // 2,)
    s[p + "head2.0.weight"] = (c.last_dim, F_ // 2, 3, 3, 3)
    s[p + "head2.0.bias"] = (c.last_dim,)
    s[p + "head2.2.weight"] = (out_ch, c.last_dim, 1, 1, 1)
    s[p + "head2.2.bias"] = (out_ch,)
    ap = actpost_of(task)
    for i in range(4):
        s[f"{p}act_postprocess.{i}.0.weight"] = (L[i], c.dim, 1, 1, 1)
        s[f"{p}act_postprocess.{i}.0.bias"] = (L[i],)
        sf = ap[i]
        if any(x > 0 for x in sf):  # ConvTranspose3d, kernel = stride = 2**s   (dpt_block.py:255-265)
            k = tuple(2 ** x for x in sf)
            s[f"{p}act_postprocess.{i}.1.weight"] = (L[i], L[i]) + k
            s[f"{p}act_postprocess.{i}.1.bias"] = (L[i],)
        elif any(x < 0 for x in sf):  # strided Conv3d  (dpt_block.py:266-276)
            stride = tuple(2 ** (-x) for x in sf)
            k = tuple((st // 2) * 2 + 1 for st in stride)
            s[f"{p}act_postprocess.{i}.1.weight"] = (L[i], L[i]) + k
            s[f"{p}act_postprocess.{i}.1.bias"] = (L[i],)
    return s


def track_schema(c: ModelCfg, task: str = "track_2d") -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    p = f"task_heads.{task}."
    D = c.dim
    s[p + "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = (3, D // 2)
    for i in range(2):
        s[f"{p}prompt_encoder.point_embeddings.{i}.weight"] = (1, D)
    for i in range(2):
        s[f"{p}prompt_encoder.prompt_feature_embeddings.{i}.weight"] = (1, D)
    s[p + "prompt_encoder.not_a_point_embed.weight"] = (1, D)
    s[p + "prompt_encoder.no_mask_embed.weight"] = (1, D)
    t = p + "mask_decoder.transformer."

    def attn(base: str, internal: int):
        for nm in ("q_proj", "k_proj", "v_proj"):
            s[f"{base}.{nm}.weight"] = (internal, D)
            s[f"{base}.{nm}.bias"] = (internal,)
        s[f"{base}.out_proj.weight"] = (D, internal)
        s[f"{base}.out_proj.bias"] = (D,)

    for l in range(c.sam_depth):
        lp = f"{t}layers.{l}."
        attn(lp + "self_attn", D)
        s[lp + "norm1.weight"] = (D,)
        s[lp + "norm1.bias"] = (D,)
        attn(lp + "cross_attn_token_to_image", D // 2)
        s[lp + "norm2.weight"] = (D,)
        s[lp + "norm2.bias"] = (D,)
        s[lp + "mlp.lin1.weight"] = (c.sam_mlp, D)
        s[lp + "mlp.lin1.bias"] = (c.sam_mlp,)
        s[lp + "mlp.lin2.weight"] = (D, c.sam_mlp)
        s[lp + "mlp.lin2.bias"] = (D,)
        s[lp + "norm3.weight"] = (D,)
        s[lp + "norm3.bias"] = (D,)
        s[lp + "norm4.weight"] = (D,)
        s[lp + "norm4.bias"] = (D,)
        attn(lp + "cross_attn_image_to_token", D // 2)
    attn(t + "final_attn_token_to_image", D // 2)
    s[t + "norm_final_attn.weight"] = (D,)
    s[t + "norm_final_attn.bias"] = (D,)
    m = p + "mask_decoder."
    s[m + "iou_token.weight"] = (1, D)
    s[m + "mask_tokens.weight"] = (3, D)
    d0 = min(2 * D // c.decoding_out_dim_factor, D)
    d1 = D // c.decoding_out_dim_factor
    s[m + "output_upscaling.0.weight"] = (D, d0, 2, 2, 2)
    s[m + "output_upscaling.0.bias"] = (d0,)
    s[m + "output_upscaling.1.weight"] = (d0,)
    s[m + "output_upscaling.1.bias"] = (d0,)
    s[m + "output_upscaling.3.weight"] = (d0, d1, 1, 2, 2)
    s[m + "output_upscaling.3.bias"] = (d1,)
    for i in range(3):
        dims = [(D, D), (D, D), (d1, D)]
        for j in range(3):
            s[f"{m}output_hypernetworks_mlps.{i}.layers.{j}.weight"] = dims[j]
            s[f"{m}output_hypernetworks_mlps.{i}.layers.{j}.bias"] = (dims[j][0],)
    s[p + "prompt_feature_linear_layer.weight"] = (D, D)
    s[p + "prompt_feature_linear_layer.bias"] = (D,)
    s[p + "processed_video_mask_token.weight"] = (1, D)
    s[p + "processed_video_features_proj.weight"] = (D, D)
    s[p + "processed_video_features_proj.bias"] = (D,)
    return s


def state_dict_schema(c: ModelCfg, tasks: Optional[List[str]] = None) -> "OrderedDict[str, Tuple[int, ...]]":
    """Keys (relative to ``l4p_model.``) and shapes of the reference L4P_VideoMAE state_dict."""
    s = encoder_schema(c)
    for task, och in c.dense_tasks.items():
        if tasks is None or task in tasks:
            s.update(dpt_schema(c, task, och))
    if tasks is None or "track_2d" in tasks:
        s.update(track_schema(c))
    return s


# ------------------------------------------------------------------------------------------------
# name-seeded values
# ------------------------------------------------------------------------------------------------
def _canonical(name: str) -> str:
    # scratch.layerN_rn is the same module object as scratch.layer_rn.{N-1}  (dpt_block.py:44-88)
    for i in range(1, 5):
        name = name.replace(f"scratch.layer{i}_rn.", f"scratch.layer_rn.{i - 1}.")
    return name


def _is_conv_transpose(name: str) -> bool:
    if "output_upscaling" in name:
        return True
    return "act_postprocess.0.1." in name or "act_postprocess.1.1." in name


def seeded_tensor(name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    """Deterministic fp32 tensor for state-dict key ``name`` (CPU generator => identical everywhere)."""
    cname = _canonical(name)
    g = torch.Generator().manual_seed(zlib.crc32(cname.encode("utf-8")))
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    leaf = cname.rsplit(".", 1)[-1]
    is_norm = ".norm" in cname or "norm_final_attn" in cname or "output_upscaling.1." in cname
    if "positional_encoding_gaussian_matrix" in cname:
        return t
    if is_norm and leaf == "weight":
        return 1.0 + 0.1 * t
    if len(shape) == 1:  # every bias, q_bias / v_bias, LayerNorm bias
        return 0.05 * t
    if len(shape) == 2 and shape[0] in (1, 3) and "embed" in cname or cname.endswith("_token.weight") or cname.endswith("mask_tokens.weight"):
        return 0.5 * t  # nn.Embedding tables
    if cname.endswith("head2.2.weight"):
        # final 1x1x1 projection: keep the logits O(1) so exp() (depth) stays well conditioned
        return t * 0.25 * (float(shape[1]) ** -0.5)
    if _is_conv_transpose(cname):
        fan_in = shape[0]
    else:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
    return t * (float(fan_in) ** -0.5)


def seeded_state_dict(c: ModelCfg, tasks: Optional[List[str]] = None) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, seeded_tensor(k, shp)) for k, shp in state_dict_schema(c, tasks).items())
