"""Repack a reference-format state_dict (models/utils.py:52-53) into the layouts
the gfx950 kernels consume, inside ONE device arena (so multi-GPU start-up is a single RCCL
broadcast of the arena over xGMI, bench.py / parallel.py).

Layouts (T = engine dtype, bf16 or f32; "rows" are padded with zeros to a multiple of 128):
  Linear [out,in]                    -> [rows(out)][in]                     T
  qkv Linear [3*H*Dh, D]             -> [rows(3*H*96)][D] (head dim 88 -> 96, zero rows)   T,  bias (q_bias,0,v_bias) f32
  PatchEmbed Conv3d [D,3,2,14,14]    -> [rows(D)][1216]  (k = ((c*2+dt)*14+dh)*14+dw, zero-padded)
  Conv3d 3x3x3 [Co,Ci,3,3,3]         -> [rows(Co)][27*Ci] (k = tap*Ci + ci, tap = (dt*3+dh)*3+dw)
  Conv3d 1x1x1 [Co,Ci,1,1,1]         -> [rows(Co)][Ci]
  ConvTranspose3d k==s [Ci,Co,k...]  -> [rows(taps*Co)][Ci] (row = tap*Co + co), bias repeated per tap
  biases, LayerNorm affine, embeddings, pos table, tiny output convs                      f32
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .weights import ModelCfg, actpost_of

DP = 96
PATCH_KP_ALIGN = 64


def _rows(w: torch.Tensor, mult: int = 128) -> torch.Tensor:
    n = w.shape[0]
    npad = (n + mult - 1) // mult * mult
    if npad == n:
        return w.contiguous()
    out = torch.zeros((npad,) + tuple(w.shape[1:]), dtype=w.dtype)
    out[:n] = w
    return out


def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """Fixed sin/cos position table, evaluated in float64 then cast (modeling_finetune.py:288-299;
    it is a plain attribute of the reference encoder, not part of the checkpoint)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    expo = 2.0 * np.floor(np.arange(d_hid, dtype=np.float64) / 2.0) / float(d_hid)
    ang = pos / np.power(10000.0, expo)[None, :]
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(ang.astype(np.float32))


def conv3_matrix(w: torch.Tensor) -> torch.Tensor:
    co, ci = w.shape[:2]
    return w.permute(0, 2, 3, 4, 1).reshape(co, 27 * ci)


def convT_matrix(w: torch.Tensor) -> torch.Tensor:
    ci, co = w.shape[:2]
    taps = w.shape[2] * w.shape[3] * w.shape[4]
    return w.permute(2, 3, 4, 1, 0).reshape(taps * co, ci)


class Packer:
    """Collects packed CPU tensors, then lays them out in one device arena."""

    def __init__(self, tdtype: torch.dtype):
        self.tdtype = tdtype
        self.items: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def T(self, name: str, w: torch.Tensor, pad_rows: bool = True):
        w = _rows(w.float()) if pad_rows else w.float().contiguous()
        self.items[name] = w.to(self.tdtype)

    def F(self, name: str, w: torch.Tensor):
        self.items[name] = w.float().contiguous()


def pack_encoder(pk: Packer, sd: Dict[str, torch.Tensor], c: ModelCfg, prefix: str = "video_encoder."):
    D, H, Dh = c.dim, c.heads, c.head_dim
    kraw = c.in_chans * c.patch[0] * c.patch[1] * c.patch[2]
    kp = (kraw + PATCH_KP_ALIGN - 1) // PATCH_KP_ALIGN * PATCH_KP_ALIGN
    w = torch.zeros(D, kp)
    w[:, :kraw] = sd[prefix + "patch_embed.proj.weight"].reshape(D, kraw)
    pk.T("enc.patch.w", w)
    pk.F("enc.patch.b", sd[prefix + "patch_embed.proj.bias"])
    pk.F("enc.pos", sinusoid_table(c.tokens, D))
    for i in range(c.depth):
        b, o = f"{prefix}blocks.{i}.", f"enc.blk{i}."
        pk.F(o + "ln1.g", sd[b + "norm1.weight"])
        pk.F(o + "ln1.b", sd[b + "norm1.bias"])
        wq = torch.zeros(3, H, DP, D)
        wq[:, :, :Dh] = sd[b + "attn.qkv.weight"].reshape(3, H, Dh, D)
        pk.T(o + "qkv.w", wq.reshape(3 * H * DP, D))
        bq = torch.zeros(3, H, DP)
        bq[0, :, :Dh] = sd[b + "attn.q_bias"].reshape(H, Dh)
        bq[2, :, :Dh] = sd[b + "attn.v_bias"].reshape(H, Dh)  # k bias is zero: modeling_finetune.py:171-175
        pk.F(o + "qkv.b", bq.reshape(-1))
        pk.T(o + "proj.w", sd[b + "attn.proj.weight"])
        pk.F(o + "proj.b", sd[b + "attn.proj.bias"])
        pk.F(o + "ln2.g", sd[b + "norm2.weight"])
        pk.F(o + "ln2.b", sd[b + "norm2.bias"])
        pk.T(o + "fc1.w", sd[b + "mlp.fc1.weight"])
        pk.F(o + "fc1.b", sd[b + "mlp.fc1.bias"])
        pk.T(o + "fc2.w", sd[b + "mlp.fc2.weight"])
        pk.F(o + "fc2.b", sd[b + "mlp.fc2.bias"])
    pk.F("enc.norm.g", sd[prefix + "norm.weight"])
    pk.F("enc.norm.b", sd[prefix + "norm.bias"])
    return kp


def pack_dpt(pk: Packer, sd: Dict[str, torch.Tensor], c: ModelCfg, task: str):
    p = f"task_heads.{task}.task_head.dpt."
    o = f"dpt.{task}."
    ap = actpost_of(task)
    for i in range(4):
        a = f"{p}act_postprocess.{i}."
        pk.T(f"{o}act{i}.0.w", sd[a + "0.weight"].reshape(sd[a + "0.weight"].shape[0], -1))
        pk.F(f"{o}act{i}.0.b", sd[a + "0.bias"])
        sf = ap[i]
        if any(x > 0 for x in sf):
            w = sd[a + "1.weight"]
            taps = w.shape[2] * w.shape[3] * w.shape[4]
            pk.T(f"{o}act{i}.1.w", convT_matrix(w))
            pk.F(f"{o}act{i}.1.b", sd[a + "1.bias"].repeat(taps))
        elif any(x < 0 for x in sf):
            pk.T(f"{o}act{i}.1.w", conv3_matrix(sd[a + "1.weight"]))
            pk.F(f"{o}act{i}.1.b", sd[a + "1.bias"])
        pk.T(f"{o}rn{i}.w", conv3_matrix(sd[f"{p}scratch.layer_rn.{i}.weight"]))
    for r in (1, 2, 3, 4):
        rp = f"{p}scratch.refinenet{r}."
        ow = sd[rp + "out_conv.weight"]
        pk.T(f"{o}ref{r}.out.w", ow.reshape(ow.shape[0], -1))
        pk.F(f"{o}ref{r}.out.b", sd[rp + "out_conv.bias"])
        for u in (1, 2):
            if r == 4 and u == 1:
                continue  # refinenet4.resConfUnit1 exists in the checkpoint but is never executed (dpt_block.py:217)
            for cv in (1, 2):
                pk.T(f"{o}ref{r}.rcu{u}.c{cv}.w", conv3_matrix(sd[f"{rp}resConfUnit{u}.conv{cv}.weight"]))
                pk.F(f"{o}ref{r}.rcu{u}.c{cv}.b", sd[f"{rp}resConfUnit{u}.conv{cv}.bias"])
    pk.T(o + "head1.w", conv3_matrix(sd[p + "head1.0.weight"]))
    pk.F(o + "head1.b", sd[p + "head1.0.bias"])
    pk.T(o + "head2.w", conv3_matrix(sd[p + "head2.0.weight"]))
    pk.F(o + "head2.b", sd[p + "head2.0.bias"])
    w = sd[p + "head2.2.weight"]
    pk.F(o + "out.w", w.reshape(w.shape[0], -1))
    pk.F(o + "out.b", sd[p + "head2.2.bias"])


def pack_track(pk: Packer, sd: Dict[str, torch.Tensor], c: ModelCfg, task: str = "track_2d"):
    p = f"task_heads.{task}."
    o = "trk."
    G = sd[p + "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"].float()
    pk.F(o + "gauss", G)
    # dense positional encoding of the (t, x, y) cell centres: a constant of G, so it is tabulated at pack time
    # exactly as PositionEmbeddingRandom3D.forward does (prompt_encoder.py:205-219) -> [P][C], token-major
    nt, nh, nw = c.grid
    ones = torch.ones((nt, nh, nw), dtype=torch.float32)
    te, ye, xe = (ones.cumsum(0) - 0.5) / nt, (ones.cumsum(1) - 0.5) / nh, (ones.cumsum(2) - 0.5) / nw
    coords = 2 * torch.stack([te, xe, ye], dim=-1) - 1
    ang = 2 * np.pi * (coords @ G)
    pk.F(o + "dense_pe", torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1).reshape(nt * nh * nw, -1))
    for i in range(2):
        pk.F(f"{o}point_emb{i}", sd[f"{p}prompt_encoder.point_embeddings.{i}.weight"].reshape(-1))
        pk.F(f"{o}feat_emb{i}", sd[f"{p}prompt_encoder.prompt_feature_embeddings.{i}.weight"].reshape(-1))
    pk.F(o + "not_a_point", sd[p + "prompt_encoder.not_a_point_embed.weight"].reshape(-1))
    pk.F(o + "mask_tokens", sd[p + "mask_decoder.mask_tokens.weight"])
    pk.F(o + "history_mask_token", sd[p + "processed_video_mask_token.weight"].reshape(-1))
    t = p + "mask_decoder.transformer."

    def attn(src: str, dst: str):
        for nm in ("q", "k", "v", "out"):
            pk.T(f"{dst}.{nm}.w", sd[f"{src}.{nm}_proj.weight"])
            pk.F(f"{dst}.{nm}.b", sd[f"{src}.{nm}_proj.bias"])

    def norm(src: str, dst: str):
        pk.F(dst + ".g", sd[src + ".weight"])
        pk.F(dst + ".b", sd[src + ".bias"])

    def fold_i2t(src: str, dst: str):
        """Image -> token attention with its image-side projections folded into the token side (sam/transformer.py:180-185,
        223-245; used from sparse_heads._window / csrc/api_trackwin.hip once every track owns its keys):
          scores[p, t, h] = scale * (kP[p] Wq^T + bq)_h . k_t,h  =  kP[p] . K'[t, h]  +  c[t, h]
          out[p]          = sum_{t,h} softmax_t(scores)[p, t, h] * (Wout_h v_t,h)  + bout  =  P[p] . V'  + bout
        K' = k_tok x qfold^T, c = k_tok x cfold^T, V' = v_tok x ofold^T are three small GEMMs on the 6 tokens of a track, with
        block-diagonal weights (head h of the token operand only meets head h's rows): qfold [(h, ch)][j] = scale * Wq[j, ch] for
        j in head h, else 0; cfold [h][j] = scale * bq[j] likewise; ofold [(h, ch)][j] = Wout[ch, j] likewise.  The image-side
        i2t.q (P x C x C/2 MACs per track) and i2t.out projections and their [P, C/2] intermediates disappear."""
        wq, bq = sd[f"{src}.q_proj.weight"].float(), sd[f"{src}.q_proj.bias"].float()
        wo = sd[f"{src}.out_proj.weight"].float()
        inner, C_ = wq.shape
        heads = c.sam_heads
        hd = inner // heads
        scale = hd ** -0.5
        qf = torch.zeros(heads * C_, inner)
        cf = torch.zeros(heads, inner)
        of = torch.zeros(heads * C_, inner)
        for h in range(heads):
            js = slice(h * hd, (h + 1) * hd)
            qf[h * C_:(h + 1) * C_, js] = scale * wq[js].t()
            cf[h, js] = scale * bq[js]
            of[h * C_:(h + 1) * C_, js] = wo[:, js]
        pk.T(f"{dst}.qfold.w", qf)
        pk.T(f"{dst}.cfold.w", cf)
        pk.T(f"{dst}.ofold.w", of)

    def fold_t2i(src: str, dst: str):
        """Token -> image attention with the keys' projection folded into the tokens (sam/transformer.py:168-173,103-109):
        scores[t, h, p] = scale * q_t,h . (kP[p] Wk^T + bk)_h = kP[p] . Q'[t, h] + const(t, h), Q' = q_tok x kfold^T with
        kfold [(h, ch)][j] = scale * Wk[j, ch] for j in head h, else 0; the constant drops out of the softmax over p."""
        wk = sd[f"{src}.k_proj.weight"].float()
        inner, C_ = wk.shape
        heads = c.sam_heads
        hd = inner // heads
        kf = torch.zeros(heads * C_, inner)
        for h in range(heads):
            js = slice(h * hd, (h + 1) * hd)
            kf[h * C_:(h + 1) * C_, js] = hd ** -0.5 * wk[js].t()
        pk.T(f"{dst}.kfold.w", kf)

    for l in range(c.sam_depth):
        lp, lo = f"{t}layers.{l}.", f"{o}l{l}."
        attn(lp + "self_attn", lo + "self")
        attn(lp + "cross_attn_token_to_image", lo + "t2i")
        attn(lp + "cross_attn_image_to_token", lo + "i2t")
        fold_i2t(lp + "cross_attn_image_to_token", lo + "i2t")
        fold_t2i(lp + "cross_attn_token_to_image", lo + "t2i")
        for k in (1, 2, 3, 4):
            norm(f"{lp}norm{k}", f"{lo}norm{k}")
        pk.T(lo + "mlp1.w", sd[lp + "mlp.lin1.weight"])
        pk.F(lo + "mlp1.b", sd[lp + "mlp.lin1.bias"])
        pk.T(lo + "mlp2.w", sd[lp + "mlp.lin2.weight"])
        pk.F(lo + "mlp2.b", sd[lp + "mlp.lin2.bias"])
    attn(t + "final_attn_token_to_image", o + "final")
    fold_t2i(t + "final_attn_token_to_image", o + "final")
    norm(t + "norm_final_attn", o + "norm_final")
    m = p + "mask_decoder."
    w0 = sd[m + "output_upscaling.0.weight"]
    pk.T(o + "up0.w", convT_matrix(w0))
    pk.F(o + "up0.b", sd[m + "output_upscaling.0.bias"].repeat(8))
    norm(m + "output_upscaling.1", o + "up_ln")
    # last up-scaling ConvTranspose (1,2,2): every tap's d1 output channels are zero-padded to a multiple of 32 so that
    # a 32-column chunk of the fused mask-product epilogue (L4P_EPI_MASKDOT) never straddles two taps
    w3 = sd[m + "output_upscaling.3.weight"]
    d1 = w3.shape[1]
    d1p = (d1 + 31) // 32 * 32
    w3m = convT_matrix(w3).reshape(4, d1, -1)
    pk.T(o + "up1.w", torch.nn.functional.pad(w3m, (0, 0, 0, d1p - d1)).reshape(4 * d1p, -1))
    pk.F(o + "up1.b", torch.nn.functional.pad(sd[m + "output_upscaling.3.bias"].float(), (0, d1p - d1)).repeat(4))
    for i in range(3):
        for j in range(3):
            pk.T(f"{o}hyper{i}.{j}.w", sd[f"{m}output_hypernetworks_mlps.{i}.layers.{j}.weight"])
            pk.F(f"{o}hyper{i}.{j}.b", sd[f"{m}output_hypernetworks_mlps.{i}.layers.{j}.bias"])
    pk.T(o + "prompt_lin.w", sd[p + "prompt_feature_linear_layer.weight"])
    pk.F(o + "prompt_lin.b", sd[p + "prompt_feature_linear_layer.bias"])
    pk.T(o + "history_proj.w", sd[p + "processed_video_features_proj.weight"])
    pk.F(o + "history_proj.b", sd[p + "processed_video_features_proj.bias"])


class PackedWeights:
    """name -> device tensor views into one contiguous arena."""

    def __init__(self, layout: List[Tuple[str, Tuple[int, ...], torch.dtype, int]], arena: torch.Tensor, meta: dict):
        self.layout = layout
        self.arena = arena
        self.meta = meta
        self.t: Dict[str, torch.Tensor] = {}
        for name, shape, dt, off in layout:
            n = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            self.t[name] = arena[off:off + n].view(dt).view(shape)

    def __getitem__(self, k: str) -> torch.Tensor:
        return self.t[k]

    def __contains__(self, k: str) -> bool:
        return k in self.t

    @staticmethod
    def from_packer(pk: Packer, device: torch.device, meta: dict) -> "PackedWeights":
        layout, off = [], 0
        for name, w in pk.items.items():
            layout.append((name, tuple(w.shape), w.dtype, off))
            off += (w.numel() * w.element_size() + 255) // 256 * 256
        arena = torch.empty(off, dtype=torch.uint8, device=device)
        pw = PackedWeights(layout, arena, meta)
        for name, w in pk.items.items():
            pw.t[name].copy_(w)
        return pw

    # ---- offline arena files (tools/ckpt_to_arena.py): the packed form of a checkpoint, ready to upload ----------
    MAGIC = b"L4PARENA1\n"

    def save(self, path: str, extra: Optional[dict] = None) -> None:
        """magic | u64 header length | JSON header {layout, meta, extra} | zero padding to 4096 | raw arena bytes."""
        import json

        names = {torch.float32: "float32", torch.bfloat16: "bfloat16", torch.float16: "float16"}
        hdr = json.dumps({"layout": [[n, list(sh), names[dt], off] for n, sh, dt, off in self.layout], "meta": self.meta,
                          "nbytes": int(self.arena.numel()), "extra": extra or {}}).encode()
        with open(path, "wb") as f:
            f.write(self.MAGIC)
            f.write(len(hdr).to_bytes(8, "little"))
            f.write(hdr)
            f.write(b"\0" * ((-f.tell()) % 4096))
            f.write(self.arena.cpu().numpy().tobytes())

    @staticmethod
    def is_arena_file(path: str) -> bool:
        try:
            with open(path, "rb") as f:
                return f.read(len(PackedWeights.MAGIC)) == PackedWeights.MAGIC
        except (OSError, TypeError):
            return False

    @staticmethod
    def load(path: str, device: torch.device) -> "PackedWeights":
        import json

        with open(path, "rb") as f:
            if f.read(len(PackedWeights.MAGIC)) != PackedWeights.MAGIC:
                raise ValueError(f"{path} is not a packed L4P arena (tools/ckpt_to_arena.py writes them)")
            n = int.from_bytes(f.read(8), "little")
            hdr = json.loads(f.read(n))
            data_off = (f.tell() + 4095) // 4096 * 4096
        dts = {"float32": torch.float32, "bfloat16": torch.bfloat16, "float16": torch.float16}
        layout = [(nm, tuple(sh), dts[dt], off) for nm, sh, dt, off in hdr["layout"]]
        raw = np.memmap(path, dtype=np.uint8, mode="r", offset=data_off, shape=(hdr["nbytes"],))
        arena = torch.from_numpy(np.array(raw)).to(device)  # (one host copy: the map is read-only)
        pw = PackedWeights(layout, arena, hdr["meta"])
        pw.extra = hdr.get("extra", {})
        return pw

    @staticmethod
    def empty_like_layout(layout, nbytes: int, device: torch.device, meta: dict) -> "PackedWeights":
        """Receiver side of the weight broadcast: same layout, uninitialised arena."""
        return PackedWeights(layout, torch.empty(nbytes, dtype=torch.uint8, device=device), meta)


def pack_state_dict(sd: Dict[str, torch.Tensor], c: ModelCfg, tdtype: torch.dtype, device: torch.device,
                    tasks: Optional[List[str]] = None) -> PackedWeights:
    """Reference state_dict (keys relative to ``l4p_model.``) -> PackedWeights on ``device``."""
    pk = Packer(tdtype)
    kp = pack_encoder(pk, sd, c)
    for task in c.dense_tasks:
        if (tasks is None or task in tasks) and f"task_heads.{task}.task_head.dpt.head1.0.weight" in sd:
            pack_dpt(pk, sd, c, task)
    if (tasks is None or "track_2d" in tasks) and "task_heads.track_2d.mask_decoder.mask_tokens.weight" in sd:
        pack_track(pk, sd, c)
    return PackedWeights.from_packer(pk, device, {"patch_kp": kp})
