"""torch.Tensor -> raw pointer plumbing over the kernel-level C ABI (include/l4p_hip.h).

PyTorch is used here only for device memory and the current HIP stream; every FLOP runs in
libl4p_hip.so.  These wrappers are what tests/ call for per-kernel parity and what the host mirror
(l4p_amd.models.*) composes.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, EPI_CONVT, EPI_DENSE, EPI_QKV, L4P_BF16, L4P_F16, L4P_F32, GemmDesc

DP = 96  # padded attention head dim used by the kernels


def torch_dtype(dtype: int) -> torch.dtype:
    return {L4P_BF16: torch.bfloat16, L4P_F16: torch.float16, L4P_F32: torch.float32}[dtype]


def code_of(t: torch.dtype) -> int:
    if t == torch.bfloat16:
        return L4P_BF16
    if t == torch.float16:
        return L4P_F16
    if t == torch.float32:
        return L4P_F32
    raise ValueError(f"unsupported engine dtype {t}")


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "engine tensors must be contiguous device tensors"
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def pad_rows(w: torch.Tensor, mult: int = 128) -> torch.Tensor:
    """Zero-pad the leading (output-feature) dim of a packed weight to a multiple of ``mult``."""
    n = w.shape[0]
    npad = (n + mult - 1) // mult * mult
    if npad == n:
        return w.contiguous()
    out = torch.zeros((npad,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[:n] = w
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, dtype: int,
              want_T: bool = True, want_f32: bool = False) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    assert x.dtype == torch.float32
    M, Cc = x.shape
    out_T = torch.empty((M, Cc), dtype=torch_dtype(dtype), device=x.device) if want_T else None
    out_f = torch.empty((M, Cc), dtype=torch.float32, device=x.device) if want_f32 else None
    lib = _lib.load()
    _lib.check(lib.l4p_layernorm(_stream(), dtype, _p(x), _p(gamma), _p(beta), eps, _p(out_T), _p(out_f), M, Cc),
               "l4p_layernorm")
    return out_T, out_f


def gemm(a: torch.Tensor, w: torch.Tensor, n: int, *, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
         res1: Optional[torch.Tensor] = None, res2: Optional[torch.Tensor] = None, res_mod: int = 0,
         out_f32: bool = False, out_T: bool = True, out: Optional[torch.Tensor] = None) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """out[m][n] = act(a @ w[:n].T + bias) + res1 + res2.  ``w`` is [ceil128(n)][K]."""
    dtype = code_of(a.dtype)
    M, K = a.shape
    assert w.shape[1] == K and w.shape[0] % 128 == 0 and w.shape[0] >= n and w.dtype == a.dtype
    d = GemmDesc()
    d.A, d.lda, d.W, d.ldw = _p(a), K, _p(w), K
    d.M, d.N, d.K = M, n, K
    d.bias = _p(bias)
    d.act = act
    if res1 is not None:
        d.res1 = _p(res1)
        d.res2 = _p(res2)
        d.res_f32 = 1 if res1.dtype == torch.float32 else 0
        d.ldr = res1.shape[-1]
        d.res_mod = res_mod
    of = torch.empty((M, n), dtype=torch.float32, device=a.device) if out_f32 else None
    oT = None
    if out is not None:
        if out.dtype == torch.float32 and dtype != L4P_F32:
            of = out
        else:
            oT = out
    elif out_T:
        oT = torch.empty((M, n), dtype=a.dtype, device=a.device)
    d.out_f32, d.out_T, d.ldc = _p(of), _p(oT), n
    d.epi = EPI_DENSE
    lib = _lib.load()
    _lib.check(lib.l4p_gemm(_stream(), dtype, C.byref(d)), "l4p_gemm")
    return oT, of


def splitk_for(M: int, N: int, K: int, es: int) -> int:
    """Split-K factor for problems whose output has too few tiles to occupy 256 CUs while K is long (the low-resolution
    DPT convs: M = 256..16384 voxels, K = 27*256..27*1024).  Mirrors csrc/api_dpt.hip:splitk_for: N >= 256 runs on 128x128
    tiles (512 workgroups aimed at), narrower outputs on 128x64 tiles (1024 workgroups)."""
    nk = (K + 128 // es - 1) // (128 // es)
    if nk < 32:
        return 1
    if N >= 256:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        if tiles >= 400:
            return 1
        s = 512 // tiles
    else:
        tiles = ((M + 127) // 128) * ((N + 63) // 64)
        if tiles >= 512:
            return 1
        s = 1024 // tiles
    return max(1, min(16, s, nk // 8))


def kv_block(dtype: torch.dtype) -> int:
    """Keys per attention KV block (a 128-byte V^T tile row): 64 for bf16, 32 for f32."""
    return 128 // torch.empty((), dtype=dtype).element_size()


def k_tile_order(k: torch.Tensor) -> torch.Tensor:
    """Host-side statement of the K tile order the kernels use (tests / documentation):
    k [B,S,H,96] -> 8-element groups [B][H][S/KVB][6][KVB][half ^ ((key>>3)&1)]."""
    B, S, H, dp = k.shape
    kvb = kv_block(k.dtype)
    g = k.view(B, S // kvb, kvb, H, dp // 16, 2, 8).permute(0, 3, 1, 4, 2, 5, 6).contiguous()  # B H kb ks key half 8
    key = torch.arange(kvb, device=k.device)
    flip = ((key >> 3) & 1).bool()
    g[:, :, :, :, flip] = g[:, :, :, :, flip].flip(-2)
    return g.reshape(-1)


def qkv_gemm(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, B: int, S: int, H: int, q_scale: float = 0.0):
    """Fused qkv projection writing the attention layouts: q [B*S, H*96], kt (tile order, flat), vt [B,H,96,S].
    ``q_scale`` != 0: the q columns are multiplied by it before rounding (the engine passes head_dim^-0.5 * log2 e and
    then calls attention(..., scale=0.0): pre-scaled)."""
    dtype = code_of(a.dtype)
    M, K = a.shape
    assert M == B * S and w.shape[0] >= 3 * H * DP and S % kv_block(a.dtype) == 0
    q = torch.empty((M, H * DP), dtype=a.dtype, device=a.device)
    kt = torch.empty((M * H * DP,), dtype=a.dtype, device=a.device)
    vt = torch.empty((B, H, DP, S), dtype=a.dtype, device=a.device)
    d = GemmDesc()
    d.A, d.lda, d.W, d.ldw = _p(a), K, _p(w), K
    d.M, d.N, d.K = M, 3 * H * DP, K
    d.bias = _p(bias)
    d.out_T, d.ldc = _p(q), H * DP
    d.epi = EPI_QKV
    d.k_tiled = _p(kt)
    d.vt, d.S, d.H, d.Dp = _p(vt), S, H, DP
    d.q_scale = q_scale
    lib = _lib.load()
    _lib.check(lib.l4p_gemm(_stream(), dtype, C.byref(d)), "l4p_gemm(qkv)")
    return q, kt, vt


def attention(q: torch.Tensor, kt: torch.Tensor, vt: torch.Tensor, head_dim: int, scale: Optional[float] = None) -> torch.Tensor:
    B, H, dp, S = vt.shape
    assert dp == DP and tuple(q.shape) == (B * S, H * DP) and kt.numel() == q.numel()
    out = torch.empty((B * S, H * head_dim), dtype=q.dtype, device=q.device)
    scale = head_dim ** -0.5 if scale is None else scale
    lib = _lib.load()
    _lib.check(lib.l4p_attention(_stream(), code_of(q.dtype), _p(q), _p(kt), _p(vt), _p(out), B, S, H, head_dim, scale),
               "l4p_attention")
    return out


def patch_gather(rgb: torch.Tensor, patch: Tuple[int, int, int], kp: int, dtype: int) -> torch.Tensor:
    B, Cin, T, H, W = rgb.shape
    pt, ph, pw = patch
    tokens = (T // pt) * (H // ph) * (W // pw)
    out = torch.empty((B * tokens, kp), dtype=torch_dtype(dtype), device=rgb.device)
    lib = _lib.load()
    _lib.check(lib.l4p_patch_gather(_stream(), dtype, _p(rgb), _p(out), B, Cin, T, H, W, pt, ph, pw, kp),
               "l4p_patch_gather")
    return out


def conv3d_k3(x: torch.Tensor, w: torch.Tensor, cout: int, *, stride: Tuple[int, int, int] = (1, 1, 1),
              bias: Optional[torch.Tensor] = None, relu_in: bool = False, act: int = ACT_NONE,
              res1: Optional[torch.Tensor] = None, res2: Optional[torch.Tensor] = None, relu_copy: bool = False,
              ups_to: Optional[Tuple[int, int]] = None):
    """3x3x3 conv, pad 1, channels-last: x [B,T,H,W,Cin] -> [B,To,Ho,Wo,cout]; w [ceil128(cout)][27*Cin].
    ``relu_copy=True`` additionally returns relu(out) (written by the same epilogue).  ``ups_to=(H, W)``: the conv reads the
    bilinear (align_corners) up-sampling of x to (H, W), formed in its loader (l4p_gemm_desc.ups_hi / ups_wi)."""
    dtype = code_of(x.dtype)
    B, Ti, Hi, Wi, Cin = x.shape
    lo = None
    if ups_to is not None:
        lo, (Hi, Wi) = (Hi, Wi), ups_to
    st, sh, sw = stride
    To, Ho, Wo = (Ti - 1) // st + 1, (Hi - 1) // sh + 1, (Wi - 1) // sw + 1
    out = torch.empty((B, To, Ho, Wo, cout), dtype=x.dtype, device=x.device)
    d = GemmDesc()
    d.A, d.W, d.ldw = _p(x), _p(w), 27 * Cin
    d.M, d.N, d.K = B * To * Ho * Wo, cout, 27 * Cin
    d.Ti, d.Hi, d.Wi, d.Cin, d.To, d.Ho, d.Wo = Ti, Hi, Wi, Cin, To, Ho, Wo
    d.st, d.sh, d.sw, d.relu_in = st, sh, sw, 1 if relu_in else 0
    d.bias, d.act = _p(bias), act
    if res1 is not None:
        d.res1, d.res2, d.res_f32, d.ldr = _p(res1), _p(res2), 0, cout
    d.out_T, d.ldc = _p(out), cout
    if lo is not None:
        d.ups_hi, d.ups_wi = lo
    out_relu = torch.empty_like(out) if relu_copy else None
    d.out_relu_T = _p(out_relu)
    sk = splitk_for(d.M, cout, 27 * Cin, x.element_size()) if lo is None else 1
    if sk > 1:
        partial = torch.empty((sk, d.M, cout), dtype=torch.float32, device=x.device)
        d.splitk, d.partial = sk, _p(partial)
    lib = _lib.load()
    _lib.check(lib.l4p_conv3d_k3(_stream(), dtype, C.byref(d)), "l4p_conv3d_k3")
    return (out, out_relu) if relu_copy else out


def conv_transpose(x: torch.Tensor, w: torch.Tensor, cout: int, k: Tuple[int, int, int],
                   bias_taps: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """ConvTranspose3d with kernel == stride == k, channels-last.  w: [ceil128(taps*cout)][Cin], row = tap*cout+co;
    bias_taps: float [taps*cout] (the conv bias repeated per tap)."""
    dtype = code_of(x.dtype)
    B, Ti, Hi, Wi, Cin = x.shape
    kt, kh, kw = k
    n = kt * kh * kw * cout
    out = torch.empty((B, Ti * kt, Hi * kh, Wi * kw, cout), dtype=x.dtype, device=x.device)
    d = GemmDesc()
    d.A, d.lda, d.W, d.ldw = _p(x), Cin, _p(w), Cin
    d.M, d.N, d.K = B * Ti * Hi * Wi, n, Cin
    d.Ti, d.Hi, d.Wi = Ti, Hi, Wi
    d.bias, d.act = _p(bias_taps), act
    d.out_T = _p(out)
    d.epi = EPI_CONVT
    d.kt, d.kh, d.kw, d.Cout = kt, kh, kw, cout
    lib = _lib.load()
    _lib.check(lib.l4p_gemm(_stream(), dtype, C.byref(d)), "l4p_gemm(convT)")
    return out


def upsample_trilinear(x: torch.Tensor, size: Tuple[int, int, int], align_corners: bool) -> torch.Tensor:
    """Channels-last trilinear resize [B,Ti,Hi,Wi,C] -> [B,To,Ho,Wo,C]."""
    B, Ti, Hi, Wi, Cc = x.shape
    To, Ho, Wo = size
    if (To, Ho, Wo) == (Ti, Hi, Wi):
        return x
    y = torch.empty((B, To, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    _lib.check(lib.l4p_upsample_trilinear(_stream(), code_of(x.dtype), _p(x), _p(y), B, Ti, Hi, Wi, To, Ho, Wo, Cc,
                                          1 if align_corners else 0), "l4p_upsample_trilinear")
    return y


def head_out(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, post_exp: bool) -> torch.Tensor:
    """x [B,T,H,W,128] T -> float [B,Cout,T,H,W] = conv1x1(x) (+exp)."""
    B, Tt, Hh, Ww, Cc = x.shape
    cout = w.shape[0]
    y = torch.empty((B, cout, Tt, Hh, Ww), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.l4p_head_out(_stream(), code_of(x.dtype), _p(x), _p(w), _p(bias), _p(y), Tt * Hh * Ww, B, Cc, cout,
                                1 if post_exp else 0), "l4p_head_out")
    return y
