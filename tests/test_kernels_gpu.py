"""Per-kernel parity of the HIP kernels (through the C ABI) against plain PyTorch fp32 CPU ops on
the SAME inputs (bf16-rounded for the bf16 mode).

Tolerances (stated once, used everywhere in this file):
  * L4P_F32 mode: max |y - ref| <= 1e-3 * max|ref|  (north_star's 1e-3 relative; measured ~1e-6)
  * L4P_BF16 mode, float outputs (f32 accumulate): same 1e-3 bound
  * L4P_BF16 mode, bf16 outputs: the result is additionally rounded to bf16 (half-ulp = 2^-9 =
    1.95e-3 relative), so the bound is rel-L2 <= 3e-3 and max error <= 1 bf16 ulp of max|ref|.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from l4p_amd import ops
from l4p_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, L4P_BF16, L4P_F16, L4P_F32

MODES = [L4P_F32, L4P_BF16, L4P_F16]


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def as_mode(x, mode):
    """Round to the engine storage type and return (device tensor, fp32 cpu view of the same values)."""
    t = x.to(ops.torch_dtype(mode))
    return t.cuda(), t.float()


def check(y, ref, mode, bf16_out):
    y = y.float().cpu()
    scale = ref.abs().max().item() + 1e-30
    err = (y - ref).abs().max().item()
    if mode == L4P_BF16 and bf16_out:
        rel_l2 = ((y - ref).norm() / (ref.norm() + 1e-30)).item()
        assert rel_l2 <= 3e-3, f"rel-L2 {rel_l2:.3e}"
        assert err <= scale * 2 ** -7, f"max err {err:.3e} vs scale {scale:.3e}"
    elif mode == L4P_F16 and bf16_out:  # output rounded to half: 11 significant bits
        rel_l2 = ((y - ref).norm() / (ref.norm() + 1e-30)).item()
        assert rel_l2 <= 4e-4, f"rel-L2 {rel_l2:.3e}"
        assert err <= scale * 2 ** -10, f"max err {err:.3e} vs scale {scale:.3e}"
    else:
        assert err <= 1e-3 * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("M,C", [(2048, 1408), (37, 352), (6, 176)])
def test_layernorm(dev, mode, M, C):
    x = rnd((M, C), 1, 3.0) + 0.5
    g, b = rnd((C,), 2) * 0.2 + 1.0, rnd((C,), 3) * 0.1
    ref = F.layer_norm(x, (C,), g, b, 1e-6)
    yT, yf = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-6, mode, want_T=True, want_f32=True)
    check(yf, ref, mode, False)
    check(yT, ref, mode, True)


@pytest.mark.parametrize("mode", [L4P_BF16, L4P_F16])
@pytest.mark.parametrize("M,C", [(8192 + 37, 352), (4096, 512), (5000, 64), (4099, 176)])
@pytest.mark.parametrize("act", [ACT_NONE, ACT_GELU])
def test_layernorm_of_engine_dtype_rows_in_place(dev, knob, mode, M, C, act):
    """l4p_layernorm_t (LayerNorm3d + GELU of the tracker's up-scaling, mask_decoder.py:145-157, on rows stored in the engine
    dtype, in place): the short-row kernel (knob "ln_rows16": one DPP row of 16 lanes per row, four rows per wave at a time,
    ragged row counts) and the one-wave-per-row kernel against torch on the same rounded rows; the two agree to an output ulp."""
    import ctypes as C_

    from l4p_amd import _lib
    from l4p_amd.ops import _p, _stream

    lib = _lib.load()
    xT, xf = as_mode(rnd((M, C), 5, 2.0) + 0.3, mode)
    g, b = rnd((C,), 6) * 0.2 + 1.0, rnd((C,), 7) * 0.1
    ref = F.layer_norm(xf, (C,), g, b, 1e-6)
    if act == ACT_GELU:
        ref = F.gelu(ref)
    outs = []
    gd, bd = g.cuda(), b.cuda()
    for rows16 in (0, 1):
        knob("ln_rows16", rows16)
        y = xT.clone()
        _lib.check(lib.l4p_layernorm_t(_stream(), mode, _p(y), _p(gd), _p(bd), C_.c_float(1e-6), _p(y), M, C, act), "l4p_layernorm_t")
        torch.cuda.synchronize()
        check(y, ref, mode, True)
        outs.append(y.float())
    ulp = 2.0 ** (-7 if mode == L4P_BF16 else -10)
    big = torch.maximum(outs[0].abs(), outs[1].abs())
    assert bool(((outs[0] - outs[1]).abs() <= 1.01 * ulp * big + 1e-4).all())  # (one output ulp where a value sits on a rounding boundary)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("M,N,K", [(2048, 1408, 1408), (2048, 6144, 1408), (2048, 1408, 6144), (300, 704, 1408),
                                     (2048, 1408, 1216), (70, 176, 352), (256, 256, 176)])
def test_gemm_bias(dev, mode, M, N, K):
    a, a_ref = as_mode(rnd((M, K), 10), mode)
    w, w_ref = as_mode(rnd((N, K), 11, K ** -0.5), mode)
    bias = rnd((N,), 12)
    ref = a_ref @ w_ref.t() + bias
    yT, yf = ops.gemm(a, ops.pad_rows(w), N, bias=bias.cuda(), out_f32=True, out_T=True)
    check(yf, ref, mode, False)
    check(yT, ref, mode, True)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("act", [ACT_GELU, ACT_RELU])
def test_gemm_act_residual(dev, mode, act):
    M, N, K = 512, 1408, 704
    a, a_ref = as_mode(rnd((M, K), 20), mode)
    w, w_ref = as_mode(rnd((N, K), 21, K ** -0.5), mode)
    bias = rnd((N,), 22)
    r1, r2 = rnd((M, N), 23), rnd((M, N), 24)
    z = a_ref @ w_ref.t() + bias
    z = F.gelu(z) if act == ACT_GELU else F.relu(z)
    # float residuals (encoder residual stream)
    _, yf = ops.gemm(a, ops.pad_rows(w), N, bias=bias.cuda(), act=act, res1=r1.cuda(), res2=r2.cuda(), out_f32=True,
                     out_T=False)
    check(yf, z + r1 + r2, mode, False)
    # T residuals (DPT skip connections)
    r1T, r1_ref = as_mode(r1, mode)
    r2T, r2_ref = as_mode(r2, mode)
    yT, _ = ops.gemm(a, ops.pad_rows(w), N, bias=bias.cuda(), act=act, res1=r1T, res2=r2T)
    check(yT, z + r1_ref + r2_ref, mode, True)


@pytest.mark.parametrize("mode", MODES)
def test_gemm_broadcast_residual(dev, mode):
    # pos-embed style: residual table indexed by m % res_mod
    M, N, K, S = 512, 352, 192, 256
    a, a_ref = as_mode(rnd((M, K), 30), mode)
    w, w_ref = as_mode(rnd((N, K), 31, K ** -0.5), mode)
    pos = rnd((S, N), 32)
    ref = a_ref @ w_ref.t() + pos.repeat(M // S, 1)
    _, yf = ops.gemm(a, ops.pad_rows(w), N, res1=pos.cuda(), res_mod=S, out_f32=True, out_T=False)
    check(yf, ref, mode, False)


LOG2E = 1.4426950408889634


def check_attn(out, ref, mode, prescaled):
    """Attention outputs.  Pre-scaled q (the engine's form): the file's bf16-output bound.  scale applied inside the bf16
    kernel: q * scale * log2(e) is rounded to bf16 a second time in registers, which perturbs every score by ~2^-9
    relative — bound 1.5x looser (rel-L2 4.5e-3, 2 ulp of the maximum)."""
    if prescaled or mode == L4P_F32:
        return check(out, ref, mode, True)
    y = out.float().cpu()
    rel_l2 = ((y - ref).norm() / (ref.norm() + 1e-30)).item()
    assert rel_l2 <= 4.5e-3, f"rel-L2 {rel_l2:.3e}"
    assert (y - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -6


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("prescaled", [True, False])
@pytest.mark.parametrize("B,S,H,Dh", [(1, 2048, 16, 88), (2, 256, 2, 88), (1, 128, 3, 64)])
def test_qkv_attention(dev, mode, prescaled, B, S, H, Dh):
    """qkv GEMM epilogue layouts (q dense — optionally pre-multiplied by head_dim^-0.5 log2 e as the engine does —, k in
    tile order, v transposed) + fused attention vs softmax(q k^T / sqrt(d)) v."""
    C = H * Dh
    x, x_ref = as_mode(rnd((B * S, C), 40), mode)
    wqkv = rnd((3 * C, C), 41, C ** -0.5)
    qb, vb = rnd((C,), 42) * 0.1, rnd((C,), 43) * 0.1
    # pack: [3][H][96][C] with zero rows for d >= Dh; bias = (q_bias, 0, v_bias)
    wp = torch.zeros(3, H, ops.DP, C)
    wp[:, :, :Dh] = wqkv.view(3, H, Dh, C)
    bp = torch.zeros(3, H, ops.DP)
    bp[0, :, :Dh] = qb.view(H, Dh)
    bp[2, :, :Dh] = vb.view(H, Dh)
    w, w_ref = as_mode(wp.view(3 * H * ops.DP, C), mode)
    qs = Dh ** -0.5 * LOG2E if prescaled else 0.0
    q, kt, vt = ops.qkv_gemm(x, ops.pad_rows(w), bp.view(-1).cuda(), B, S, H, q_scale=qs)
    full = (x_ref @ w_ref.t() + bp.view(-1)).view(B, S, 3, H, ops.DP)
    check(q.view(B, S, H, ops.DP), full[:, :, 0] * (qs if prescaled else 1.0), mode, True)
    check(vt, full[:, :, 2].permute(0, 2, 3, 1), mode, True)
    k_expect = ops.k_tile_order(full[:, :, 1].contiguous().to(ops.torch_dtype(mode))).float()
    check(kt, k_expect, mode, True)
    # attention on the values the kernel actually produced (rounded to T)
    q_ref = q.float().cpu().view(B, S, H, ops.DP).permute(0, 2, 1, 3)  # B H S 96
    k_ref = full[:, :, 1].to(ops.torch_dtype(mode)).float().permute(0, 2, 1, 3)
    v_ref = vt.float().cpu().permute(0, 1, 3, 2)  # B H S 96
    # pre-scaled q holds q * scale * log2(e): the weights are exp2(q' k^T) = exp(ln 2 * q' k^T)
    attn = torch.softmax((q_ref * (math.log(2.0) if prescaled else Dh ** -0.5)) @ k_ref.transpose(-2, -1), dim=-1)
    ref = (attn @ v_ref)[..., :Dh].transpose(1, 2).reshape(B * S, C)
    out = ops.attention(q, kt, vt, Dh, scale=0.0 if prescaled else None)
    check_attn(out, ref, mode, prescaled)


def _attn_inputs(q4, k4, v4, mode):
    """q4,k4,v4: [B,S,H,96] float -> device tensors in the kernel layouts + fp32 views of the rounded values."""
    td = ops.torch_dtype(mode)
    B, S, H, dp = q4.shape
    qT, kT, vT = q4.to(td), k4.to(td), v4.to(td)
    q = qT.reshape(B * S, H * dp).cuda()
    kt = ops.k_tile_order(kT.cuda())
    vt = vT.permute(0, 2, 3, 1).contiguous().cuda()
    return q, kt, vt, qT.float(), kT.float(), vT.float()


def _peaked(S, seed=5, B=1, H=2, Dh=88):
    g = torch.Generator().manual_seed(seed)
    q4 = torch.randn(B, S, H, ops.DP, generator=g)
    k4 = torch.randn(B, S, H, ops.DP, generator=g)
    v4 = torch.randn(B, S, H, ops.DP, generator=g)
    for t in (q4, k4, v4):
        t[..., Dh:] = 0
    k4[0, 300:310] *= 6.0  # a few keys late in the sequence dominate
    q4[0, 17] *= 8.0
    k4[0, :64] *= 0.01     # first block nearly flat: later blocks all raise the max
    return q4, k4, v4


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("prescaled", [True, False])
@pytest.mark.parametrize("S,B,H", [(512, 1, 2), (2048, 4, 16)])  # KV-split (few workgroups) and un-split kernel forms
def test_attention_peaked_softmax(dev, mode, prescaled, S, B, H):
    """Force large running-max jumps across KV tiles: the deferred-maximum path (reference moved only when a row's block
    maximum exceeds it by 2^8, rescale of O, scores shifted in place) and its wave-uniform skip; a first block whose
    scores are far BELOW the later ones, and rows (query 17) whose scores span +-200 in the exp2 domain."""
    Dh = 88
    q4, k4, v4 = _peaked(S, B=B, H=H)
    if prescaled:
        q4 = q4 * (Dh ** -0.5 * LOG2E)
    q, kt, vt, qf, kf, vf = _attn_inputs(q4, k4, v4, mode)
    qh, kh, vh = (t.permute(0, 2, 1, 3).double() for t in (qf, kf, vf))
    attn = torch.softmax((qh * (math.log(2.0) if prescaled else Dh ** -0.5)) @ kh.transpose(-2, -1), dim=-1)
    ref = (attn @ vh)[..., :Dh].transpose(1, 2).reshape(B * S, H * Dh).float()
    out = ops.attention(q, kt, vt, Dh, scale=0.0 if prescaled else None)
    check_attn(out, ref, mode, prescaled)


@pytest.mark.parametrize("mode", [m for m in MODES if m != L4P_F32])
@pytest.mark.parametrize("B,S,H,Dh", [(2, 2048, 16, 88),   # 256 tiles: one per workgroup, no tile seam
                                      (3, 2048, 16, 88),   # 384 tiles: workgroups with one and with two tiles
                                      (11, 1024, 6, 88),   # 264 tiles of 16 KV blocks, batch * heads = 66 not a multiple of 8 (plain tile order)
                                      (4, 2048, 16, 64),   # head dim 64: the constant pieces sit elsewhere in the tile
                                      (16, 256, 16, 88)])  # the shortest sequence the kernel takes: 4 KV blocks, every step is a first / last one
def test_attention64_forms_and_the_8_wave_kernel(dev, knob, mode, B, S, H, Dh):
    """csrc/attention64.hip (one wave per SIMD, 64 query rows per wave; launches of >= 256 tiles of 256 rows) against fp32 softmax
    on the same rounded operands, in its tile-walk variants, and against the 8-wave kernel on the same inputs (knob attn64 = 0):
    the two agree to the rounding of P - they move the deferred maximum per 64 / per 32 rows."""
    g = torch.Generator().manual_seed(B * 1000 + S + Dh)
    q4 = torch.randn(B, S, H, ops.DP, generator=g) * (Dh ** -0.5 * LOG2E)
    k4 = torch.randn(B, S, H, ops.DP, generator=g)
    v4 = torch.randn(B, S, H, ops.DP, generator=g)
    for t in (q4, k4, v4):
        t[..., Dh:] = 0
    k4[1, S - 120:S - 112] *= 5.0  # a few dominant keys in one batch item: the rescale path runs in some waves and not in others
    q, kt, vt, qf, kf, vf = _attn_inputs(q4, k4, v4, mode)
    out = {}
    for v in (1, 0):
        knob("attn64", v)
        out[v] = ops.attention(q, kt, vt, Dh, scale=0.0)
        torch.cuda.synchronize()
    td = ops.torch_dtype(mode)
    ref = torch.empty(B, S, H, Dh)
    for b in range(B):  # (per batch item: the fp32 score matrix of one item is 16 x 2048 x 2048 floats)
        qh, kh, vh = (t[b].permute(1, 0, 2).double() for t in (qf, kf, vf))
        attn = torch.softmax((qh * math.log(2.0)) @ kh.transpose(-2, -1), dim=-1)
        ref[b] = (attn @ vh)[..., :Dh].permute(1, 0, 2).float()
    ref = ref.reshape(B * S, H * Dh)
    for v in (1, 0):
        check_attn(out[v], ref, mode, True)
    d = (out[1].float() - out[0].float()).abs().max().item()
    assert d <= ref.abs().max().item() * 2 ** (-6 if td == torch.bfloat16 else -9), d


@pytest.mark.parametrize("mode", MODES)
def test_attention_very_negative_and_huge_scores(dev, mode):
    """All scores of a row far below zero (the first block must SET the reference, not clamp it at 0) and a row whose
    maximum sits at +3000 in the exp2 domain (the reference is carried as two bf16 halves: 16 mantissa bits)."""
    B, S, H, Dh = 1, 256, 2, 88
    g = torch.Generator().manual_seed(11)
    q4 = torch.randn(B, S, H, ops.DP, generator=g)
    k4 = torch.randn(B, S, H, ops.DP, generator=g)
    v4 = torch.randn(B, S, H, ops.DP, generator=g)
    for t in (q4, k4, v4):
        t[..., Dh:] = 0
    q4[0, 3, :, :Dh] = -4.0
    k4[0, :, :, :Dh] = k4[0, :, :, :Dh].abs() * 0.5 + 1.0   # every key has positive entries: row 3 scores ~ -4 * 88 * 1.4
    q4[0, 5, :, :Dh] = 6.0                                    # row 5: scores ~ +6 * 88 * 1.4 * 0.1066 * 1.44 ~ +100 .. +120
    q4[0, 7, :, :Dh] = 200.0                                  # row 7: ~ +3000 and up
    q, kt, vt, qf, kf, vf = _attn_inputs(q4, k4, v4, mode)
    qh, kh, vh = (t.permute(0, 2, 1, 3).double() for t in (qf, kf, vf))
    attn = torch.softmax((qh * Dh ** -0.5) @ kh.transpose(-2, -1), dim=-1)
    ref = (attn @ vh)[..., :Dh].transpose(1, 2).reshape(B * S, H * Dh).float()
    out = ops.attention(q, kt, vt, Dh)
    assert bool(torch.isfinite(out.float()).all())
    check_attn(out, ref, mode, False)


@pytest.mark.parametrize("mode", MODES)
def test_patch_embed(dev, mode):
    """gather + GEMM == Conv3d(kernel=stride=(2,14,14)) + flatten/transpose (modeling_finetune.py:269-283)."""
    B, C = 1, 352
    rgb = rnd((B, 3, 16, 224, 224), 50)
    w = rnd((C, 3, 2, 14, 14), 51, 1176 ** -0.5)
    bias = rnd((C,), 52)
    kp = 1216
    a = ops.patch_gather(rgb.cuda(), (2, 14, 14), kp, mode)
    wp = torch.zeros(C, kp)
    wp[:, :1176] = w.view(C, 1176)
    wT, w_ref = as_mode(wp, mode)
    rgb_ref = rgb.to(ops.torch_dtype(mode)).float()
    ref = F.conv3d(rgb_ref, w_ref[:, :1176].view(C, 3, 2, 14, 14), bias, stride=(2, 14, 14)).flatten(2).transpose(1, 2)
    _, yf = ops.gemm(a, ops.pad_rows(wT), C, bias=bias.cuda(), out_f32=True, out_T=False)
    check(yf, ref.reshape(-1, C), mode, False)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape,cout,stride,relu_in", [((1, 4, 8, 8, 128), 256, (1, 1, 1), False),
                                                      ((2, 3, 10, 6, 64), 128, (1, 1, 1), True),
                                                      ((1, 8, 16, 16, 128), 128, (2, 2, 2), False)])
def test_conv3d_k3(dev, mode, shape, cout, stride, relu_in):
    B, T, H, W, Cin = shape
    x, x_ref = as_mode(rnd(shape, 60), mode)
    w = rnd((cout, Cin, 3, 3, 3), 61, (27 * Cin) ** -0.5)
    bias = rnd((cout,), 62)
    wT, w_ref = as_mode(w.permute(0, 2, 3, 4, 1).reshape(cout, 27 * Cin), mode)
    w5 = w_ref.view(cout, 3, 3, 3, Cin).permute(0, 4, 1, 2, 3)
    xin = x_ref.permute(0, 4, 1, 2, 3)
    ref = F.conv3d(F.relu(xin) if relu_in else xin, w5, bias, stride=stride, padding=1)
    skip = rnd(tuple(ref.permute(0, 2, 3, 4, 1).shape), 63)
    sT, s_ref = as_mode(skip, mode)
    y = ops.conv3d_k3(x, ops.pad_rows(wT), cout, stride=stride, bias=bias.cuda(), relu_in=relu_in, act=ACT_RELU,
                      res1=sT)
    check(y, F.relu(ref).permute(0, 2, 3, 4, 1) + s_ref, mode, True)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("k", [(2, 4, 4), (2, 2, 2), (1, 2, 2), (2, 1, 1)])
def test_conv_transpose(dev, mode, k):
    B, T, H, W, Cin, cout = 1, 4, 8, 8, 192, 64
    x, x_ref = as_mode(rnd((B, T, H, W, Cin), 70), mode)
    w = rnd((Cin, cout) + k, 71, Cin ** -0.5)  # ConvTranspose3d layout [Cin][Cout][kt][kh][kw]
    bias = rnd((cout,), 72)
    taps = k[0] * k[1] * k[2]
    wT, w_ref = as_mode(w.permute(2, 3, 4, 1, 0).reshape(taps * cout, Cin), mode)
    w5 = w_ref.view(k[0], k[1], k[2], cout, Cin).permute(4, 3, 0, 1, 2)
    ref = F.conv_transpose3d(x_ref.permute(0, 4, 1, 2, 3), w5, bias, stride=k).permute(0, 2, 3, 4, 1)
    y = ops.conv_transpose(x, ops.pad_rows(wT), cout, k, bias_taps=bias.repeat(taps).cuda())
    check(y, ref, mode, True)


@pytest.mark.parametrize("mode", MODES)
def test_conv3d_relu_copy_and_kpadded_gemm(dev, mode):
    """LDS-DMA path extras: second relu(out) output of the epilogue; K tails that are not a multiple of the k-tile."""
    B, T, H, W, Cin, cout = 1, 4, 12, 12, 128, 256
    x, x_ref = as_mode(rnd((B, T, H, W, Cin), 80), mode)
    w = rnd((cout, Cin, 3, 3, 3), 81, (27 * Cin) ** -0.5)
    wT, w_ref = as_mode(w.permute(0, 2, 3, 4, 1).reshape(cout, 27 * Cin), mode)
    ref = F.conv3d(x_ref.permute(0, 4, 1, 2, 3), w_ref.view(cout, 3, 3, 3, Cin).permute(0, 4, 1, 2, 3), padding=1).permute(0, 2, 3, 4, 1)
    y, yr = ops.conv3d_k3(x, ops.pad_rows(wT), cout, relu_copy=True)
    check(y, ref, mode, True)
    check(yr, F.relu(ref), mode, True)
    # dense GEMM with K = 352 (5.5 k-tiles of 64) and K = 88
    for K in (352, 88):
        a, a_ref = as_mode(rnd((300, K), 82), mode)
        wk, wk_ref = as_mode(rnd((176, K), 83, K ** -0.5), mode)
        yT, _ = ops.gemm(a, ops.pad_rows(wk), 176)
        check(yT, a_ref @ wk_ref.t(), mode, True)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("Nq,T,h,w,Cin,d1", [(3, 2, 8, 8, 96, 40), (2, 4, 16, 16, 352, 176)])
def test_maskdot_fused_upscaling(dev, mode, Nq, T, h, w, Cin, d1):
    """L4P_EPI_MASKDOT + l4p_mask_gather (mask_decoder.py:136-139): GELU(ConvTranspose3d(k = s = (1,2,2))) contracted with
    the per-query hyper-network vectors, against the unfused statement on the same (storage-rounded) inputs.  d1 = 40
    exercises the per-tap zero padding to a multiple of 32; the second shape is the full model's geometry."""
    import ctypes as C

    from l4p_amd import _lib
    from l4p_amd._lib import EPI_MASKDOT, GemmDesc

    x, x_ref = as_mode(rnd((Nq, T, h, w, Cin), 90), mode)
    wt = rnd((Cin, d1, 1, 2, 2), 91, Cin ** -0.5)  # ConvTranspose3d layout
    bias = rnd((d1,), 92)
    hyper = rnd((Nq, 3, d1), 93)
    d1p = (d1 + 31) // 32 * 32
    wm = wt.permute(2, 3, 4, 1, 0).reshape(4, d1, Cin)
    wp, w_ref = as_mode(F.pad(wm, (0, 0, 0, d1p - d1)).reshape(4 * d1p, Cin), mode)
    up = F.gelu(F.conv_transpose3d(x_ref.permute(0, 4, 1, 2, 3), w_ref.view(4, d1p, Cin)[:, :d1].reshape(1, 2, 2, d1, Cin)
                                   .permute(4, 3, 0, 1, 2), bias, stride=(1, 2, 2)))  # [Nq, d1, T, 2h, 2w]
    ref = torch.einsum("nic,nctyx->nityx", hyper, up)
    M = Nq * T * h * w
    cpt = d1p // 32
    hp = torch.zeros(Nq, 3, d1p)
    hp[..., :d1] = hyper
    hp = hp.cuda()
    bp = F.pad(bias, (0, d1p - d1)).repeat(4).cuda()
    partial = torch.empty(4 * cpt, 3, M, dtype=torch.float32, device="cuda")
    wpad = ops.pad_rows(wp, 256)
    d = GemmDesc()
    d.A, d.lda, d.W, d.ldw = x.data_ptr(), Cin, wpad.data_ptr(), Cin
    d.M, d.N, d.K = M, 4 * d1p, Cin
    d.bias, d.act = bp.data_ptr(), ACT_GELU
    d.out_f32 = partial.data_ptr()
    d.epi, d.Cout = EPI_MASKDOT, d1p
    d.hyper, d.hyper_rows = hp.data_ptr(), M // Nq
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.l4p_gemm(st, mode, C.byref(d)), "l4p_gemm(maskdot)")
    masks = torch.empty(Nq, 3, T, 2 * h, 2 * w, dtype=torch.float32, device="cuda")
    _lib.check(lib.l4p_mask_gather(st, partial.data_ptr(), masks.data_ptr(), Nq, T, h, w, cpt), "l4p_mask_gather")
    torch.cuda.synchronize()
    # bf16 mode: the fused path keeps the activation in float (the unfused engine path rounds it to bf16), so it is the
    # more accurate of the two; the bf16 GELU is the A&S erfc form (1.5e-7 absolute)
    check(masks, ref, mode, False)


def test_gemm_group_equals_separate_launches(dev):
    """l4p_gemm_group: up to four INDEPENDENT small bf16 GEMMs as one kernel launch (the tracker's token-side projections:
    q / k / v of the self-attention, M = 6 x tracks rows; hyper-network stages with strided A rows and a float output) —
    bit-identical to separate l4p_gemm calls; one profiler entry tagged "group"; a member that does not fit the small-problem
    kernel (here: a large M) makes the call fall back to separate launches."""
    import ctypes as C

    from l4p_amd import _lib
    from l4p_amd._lib import EPI_DENSE, GemmDesc

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    M, K = 384, 1408
    xs = [as_mode(rnd((M, K), 300 + i), L4P_BF16)[0] for i in range(2)]
    hs = as_mode(rnd((64, 6 * K), 310), L4P_BF16)[0]                       # hyper-network input: token i of a [64, 6, K] block
    ws = [ops.pad_rows(as_mode(rnd((n, K), 320 + i, K ** -0.5), L4P_BF16)[0], 128) for i, n in enumerate((1408, 1408, 704, 176))]
    bs = [rnd((n,), 330 + i).cuda() for i, n in enumerate((1408, 1408, 704, 176))]

    def descs(outs):
        ds = (GemmDesc * 4)()
        spec = [(xs[0], M, K, 1408, ACT_NONE), (xs[0], M, K, 1408, ACT_RELU), (xs[1], M, K, 704, ACT_NONE), (hs, 64, 6 * K, 176, ACT_NONE)]
        for i, (a, m, lda, n, act) in enumerate(spec):
            d = ds[i]
            d.A, d.lda, d.W, d.ldw = a.data_ptr() + (2 * K * 2 if i == 3 else 0), lda, ws[i].data_ptr(), K
            d.M, d.N, d.K = m, n, K
            d.bias, d.act, d.epi = bs[i].data_ptr(), act, EPI_DENSE
            if i == 3:
                d.out_f32, d.ldc = outs[i].data_ptr(), n
            else:
                d.out_T, d.ldc = outs[i].data_ptr(), n
        return ds

    def outs():
        return [torch.zeros(384, 1408, dtype=torch.bfloat16, device="cuda"), torch.zeros(384, 1408, dtype=torch.bfloat16, device="cuda"),
                torch.zeros(384, 704, dtype=torch.bfloat16, device="cuda"), torch.zeros(64, 176, dtype=torch.float32, device="cuda")]

    a, b = outs(), outs()
    da, db = descs(a), descs(b)
    from tests.test_gemm8p_gpu import prof_tags
    with prof_tags() as p:
        _lib.check(lib.l4p_gemm_group(st, L4P_BF16, da, 4), "l4p_gemm_group")
    tags = [ln[1] for ln in p.lines if ln[0] in ("gemm", "gemm_small")]  # (384-row problems: the small-products class)
    assert len(tags) == 1 and tags[0].startswith("group of 4"), tags
    for i in range(4):
        _lib.check(lib.l4p_gemm(st, L4P_BF16, C.byref(db[i])), "l4p_gemm")
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    ref = hs.float().cpu().reshape(64, 6, K)[:, 2] @ ws[3][:176].float().cpu().t() + bs[3].cpu()
    check(a[3], ref, L4P_BF16, False)
    # a large member: the group falls back to separate launches (still correct)
    big = as_mode(rnd((65536, K), 340), L4P_BF16)[0]
    ob = torch.zeros(65536, 1408, dtype=torch.bfloat16, device="cuda")
    ds = (GemmDesc * 2)()
    for i, (aa, m, o) in enumerate(((big, 65536, ob), (xs[0], M, a[0]))):
        ds[i].A, ds[i].lda, ds[i].W, ds[i].ldw = aa.data_ptr(), K, ws[0].data_ptr(), K
        ds[i].M, ds[i].N, ds[i].K = m, 1408, K
        ds[i].bias, ds[i].epi, ds[i].out_T, ds[i].ldc = bs[0].data_ptr(), EPI_DENSE, o.data_ptr(), 1408
    with prof_tags() as p:
        _lib.check(lib.l4p_gemm_group(st, L4P_BF16, ds, 2), "l4p_gemm_group")
    assert len([ln for ln in p.lines if ln[0] in ("gemm", "gemm_small")]) == 2  # (65536 rows: GEMM class; 384 rows: small products)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ob.float()).all()) and float(ob.float().abs().max()) > 0


@pytest.mark.parametrize("mode", [L4P_BF16, L4P_F16])
def test_gemm_skinny_equals_the_staged_kernels_bitwise(dev, knob, mode):
    """gemm_skinny.hpp (M <= 128 rows and at most 512 output blocks: one wave per 16 x 32 output block, operands streamed into fragment registers with hand-counted
    waits) against the LDS-staged kernels it replaces for the tracker's token-side projections: bit for bit - same MFMA, same k
    order, same epilogue - over the three unrolled contraction lengths and the generic loop, ragged M and N, bias / ReLU / GELU /
    float residual / both outputs, a strided A (the hyper-network's token rows), and a grouped launch; and against fp32."""
    import ctypes as C

    from l4p_amd import _lib
    from l4p_amd._lib import EPI_DENSE, GemmDesc
    from tests.test_gemm8p_gpu import prof_tags

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    dt = ops.torch_dtype(mode)

    def run(M, N, K, act, res, both, seed, lda=None, skinny=1, want_skinny=True):
        knob("gemm_skinny", skinny)
        lda = lda or K
        a = as_mode(rnd((M, lda), seed), mode)[0]
        w = ops.pad_rows(as_mode(rnd((N, K), seed + 1, K ** -0.5), mode)[0], 128)
        bias = rnd((N,), seed + 2).cuda()
        r = rnd((M, N), seed + 3).cuda() if res else None
        oT = torch.zeros(M, N, dtype=dt, device="cuda")
        of = torch.zeros(M, N, dtype=torch.float32, device="cuda") if (both or res) else None
        d = GemmDesc()
        d.A, d.lda, d.W, d.ldw = a.data_ptr(), lda, w.data_ptr(), K
        d.M, d.N, d.K = M, N, K
        d.bias, d.act, d.epi = bias.data_ptr(), act, EPI_DENSE
        if res:
            d.res1, d.res_f32, d.ldr = r.data_ptr(), 1, N
        d.out_T, d.ldc = oT.data_ptr(), N
        if of is not None:
            d.out_f32 = of.data_ptr()
        with prof_tags() as p:
            _lib.check(lib.l4p_gemm(st, mode, C.byref(d)), "l4p_gemm")
        tags = [ln[1] for ln in p.lines if ln[0] in ("gemm", "gemm_small")]
        assert len(tags) == 1 and tags[0].endswith("skinny") == bool(skinny and want_skinny), (tags, skinny)
        ref = a.float().cpu()[:, :K] @ w[:N].float().cpu().t() + bias.cpu()
        if act == ACT_RELU:
            ref = ref.clamp_min(0)
        elif act == ACT_GELU:
            ref = F.gelu(ref)
        if res:
            ref = ref + r.cpu()
        return oT, of, ref

    cases = [(48, 1408, 1408, ACT_NONE, True, False), (48, 704, 1408, ACT_NONE, False, False), (48, 2048, 1408, ACT_RELU, False, False),
             (48, 1408, 2048, ACT_NONE, True, True), (48, 1408, 704, ACT_NONE, True, False), (48, 4096, 704, ACT_NONE, False, False),
             (8, 1408, 1408, ACT_RELU, False, False), (8, 176, 1408, ACT_NONE, False, True), (48, 8, 704, ACT_NONE, False, True),
             (5, 40, 128, ACT_GELU, False, True), (64, 1408, 1344, ACT_NONE, True, False), (33, 200, 2816, ACT_RELU, False, False),
             (17, 96, 64, ACT_NONE, False, True), (96, 1408, 1408, ACT_NONE, True, False), (128, 704, 2048, ACT_RELU, False, True)]
    for i, (M, N, K, act, res, both) in enumerate(cases):
        yT, yf, ref = run(M, N, K, act, res, both, 700 + 10 * i, skinny=1)
        zT, zf, _ = run(M, N, K, act, res, both, 700 + 10 * i, skinny=0)
        assert torch.equal(yT, zT), (M, N, K, act, res)
        if yf is not None:
            assert torch.equal(yf, zf), (M, N, K, act, res)
            check(yf, ref, mode, False)
        check(yT, ref, mode, True)
    # more than 512 output blocks (the folded key products, N = 11264): the staged kernel - every row block would stream the weights again
    run(48, 11264, 704, ACT_NONE, False, False, 880, skinny=1, want_skinny=False)
    # strided A: token 2 of a [8, 6, K] block (the hyper-network MLP's input rows)
    yT, _, ref = run(8, 1408, 1408, ACT_RELU, False, False, 900, lda=6 * 1408, skinny=1)
    zT, _, _ = run(8, 1408, 1408, ACT_RELU, False, False, 900, lda=6 * 1408, skinny=0)
    assert torch.equal(yT, zT)
    check(yT, ref, mode, True)
    # grouped launch of three skinny problems (self-attention q / k / v of 8 tracks) == three staged launches
    K = 1408
    xs = [as_mode(rnd((48, K), 950 + i), mode)[0] for i in range(2)]
    ws = [ops.pad_rows(as_mode(rnd((n, K), 960 + i, K ** -0.5), mode)[0], 128) for i, n in enumerate((1408, 1408, 704))]
    bs = [rnd((n,), 970 + i).cuda() for i, n in enumerate((1408, 1408, 704))]

    def group(skinny, grouped):
        knob("gemm_skinny", skinny)
        outs = [torch.zeros(48, n, dtype=dt, device="cuda") for n in (1408, 1408, 704)]
        ds = (GemmDesc * 3)()
        for i, (a, n) in enumerate(((xs[0], 1408), (xs[0], 1408), (xs[1], 704))):
            d = ds[i]
            d.A, d.lda, d.W, d.ldw = a.data_ptr(), K, ws[i].data_ptr(), K
            d.M, d.N, d.K = 48, n, K
            d.bias, d.act, d.epi = bs[i].data_ptr(), ACT_NONE, EPI_DENSE
            d.out_T, d.ldc = outs[i].data_ptr(), n
        with prof_tags() as p:
            if grouped:
                _lib.check(lib.l4p_gemm_group(st, mode, ds, 3), "l4p_gemm_group")
            else:
                for i in range(3):
                    _lib.check(lib.l4p_gemm(st, mode, C.byref(ds[i])), "l4p_gemm")
        tags = [ln[1] for ln in p.lines if ln[0] in ("gemm", "gemm_small")]
        return outs, tags

    a, ta = group(1, True)
    b, tb = group(0, False)
    assert len(ta) == 1 and ta[0].startswith("group of 3") and ta[0].endswith("skinny"), ta
    assert len(tb) == 2 and not any(t.endswith("skinny") for t in tb), tb  # (the table is keyed by tag: two of the three share one)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("N,D,heads", [(8, 1408, 8), (3, 128, 2), (2, 2048, 8)])
def test_small_attn_kind0_vs_torch(dev, mode, N, D, heads):
    """l4p_small_attn kind 0 (sam/transformer.py:223-245 on the 6 prompt tokens of each track) against torch in fp32: the form with
    all of a lane's operands requested up front (head dims up to 192: the tracker's 176) and the rolled form (256)."""
    import ctypes as C

    from l4p_amd import _lib

    q, qr = as_mode(rnd((N, 6, D), 1200), mode)
    k, kr = as_mode(rnd((N, 6, D), 1201), mode)
    v, vr = as_mode(rnd((N, 6, D), 1202), mode)
    out = torch.empty_like(q)
    _lib.check(_lib.load().l4p_small_attn(torch.cuda.current_stream().cuda_stream, mode, 0, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                          out.data_ptr(), N, 6, D, heads), "l4p_small_attn")
    torch.cuda.synchronize()
    hd = D // heads
    sp = lambda t: t.reshape(N, 6, heads, hd).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(qr), sp(kr), sp(vr)).transpose(1, 2).reshape(N, 6, D)
    check(out, ref, mode, True)


def test_track_readout_wide_equals_the_column_kernel_bitwise(dev, knob):
    """l4p_track_readout (sparse_heads.py:572-589,645-647): the 1024-thread form (knob readout_wide: samples of 32 rows formed by all
    threads, summed per column in the original order) == the 256-thread column kernel bit for bit, and both against torch
    (trilinear resize, soft-argmax over pixel centres, spatial means)."""
    from l4p_amd import _lib

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    N, T, h, w, H, W = 8, 16, 56, 56, 224, 224
    masks = (rnd((N, 3, T, h, w), 1300) * 3).cuda()
    outs = {}
    for wide in (1, 0):
        knob("readout_wide", wide)
        traj = torch.zeros(N, 2, T, device="cuda")
        vis = torch.zeros(N, T, device="cuda")
        dep = torch.zeros(N, T, device="cuda")
        _lib.check(lib.l4p_track_readout(st, masks.data_ptr(), traj.data_ptr(), vis.data_ptr(), dep.data_ptr(), N, T, h, w, H, W),
                   "l4p_track_readout")
        torch.cuda.synchronize()
        outs[wide] = (traj, vis, dep)
    for a, b in zip(outs[1], outs[0]):
        assert torch.equal(a, b)
    up = F.interpolate(masks.cpu().double(), size=(T, H, W), mode="trilinear", align_corners=False)
    p = torch.softmax(up[:, 0].reshape(N, T, H * W), dim=-1).reshape(N, T, H, W)
    xs = torch.arange(W, dtype=torch.float64) + 0.5
    ys = torch.arange(H, dtype=torch.float64) + 0.5
    ref_traj = torch.stack([(p.sum(2) * xs).sum(-1), (p.sum(3) * ys).sum(-1)], dim=1)
    traj, vis, dep = (t.double().cpu() for t in outs[1])
    assert (traj - ref_traj).abs().max() <= 1e-3
    assert (vis - up[:, 1].mean((-1, -2))).abs().max() <= 1e-5
    assert ((dep - up[:, 2].mean((-1, -2)).exp()).abs() / dep.abs()).max() <= 1e-5


@pytest.mark.parametrize("mode", [L4P_BF16, L4P_F16])
def test_row_grouped_scores_deep_ring_equals_two_stages_bitwise(dev, knob, mode):
    """The folded score product of a few tracks (rows [g P, (g + 1) P) of the keys against track g's 48 folded rows: l4p_gemm_desc.w_gr;
    sam/transformer.py:223-245 with the projections folded, packing.py) leaves CUs idle: the four-stage form of the row-grouped
    kernel (knob track_deep) == the two-stage one bit for bit, and both against fp32."""
    import ctypes as C

    from l4p_amd import _lib
    from l4p_amd._lib import EPI_DENSE, GemmDesc

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    G, P, HT, K = 3, 2048, 48, 1408
    a, ar = as_mode(rnd((G * P, K), 1400), mode)
    w, wr = as_mode(rnd((G * HT + 128, K), 1401, K ** -0.5), mode)
    outs = []
    for deep in (1, 0):
        knob("track_deep", deep)
        o = torch.zeros(G * P, HT, device="cuda")
        d = GemmDesc()
        d.A, d.lda, d.W, d.ldw = a.data_ptr(), K, w.data_ptr(), K
        d.M, d.N, d.K = G * P, HT, K
        d.out_f32, d.ldc, d.epi = o.data_ptr(), HT, EPI_DENSE
        d.w_gr, d.w_gs, d.b_gs = P, HT * K, 0
        _lib.check(lib.l4p_gemm(st, mode, C.byref(d)), "l4p_gemm(w_gr)")
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    ref = torch.cat([ar[g * P:(g + 1) * P] @ wr[g * HT:(g + 1) * HT].t() for g in range(G)])
    check(outs[0], ref, mode, False)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("M", [48, 384])
def test_gemm_block_structured_weights_k_window_bitwise(dev, mode, M):
    """l4p_gemm_desc.kw_cols / kw_len: the tracker's folded projections (packing.py fold_i2t / fold_t2i: head h's C output columns meet head
    h's 88 inputs only, the weight is zero elsewhere) walk only the k-tiles of a tile's head - the same result bit for bit as the full
    contraction, for one launch and for the grouped launch (qfold, ofold, cfold), in the three engines."""
    import ctypes as C

    from l4p_amd import _lib
    from l4p_amd._lib import EPI_DENSE, GemmDesc

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    dt = ops.torch_dtype(mode)
    heads, hd, Cc = 8, 88, 1408
    K, N = heads * hd, heads * Cc
    a = as_mode(rnd((M, K), 1500), mode)[0]

    def folded(seed):
        w = torch.zeros(N, K)
        blk = rnd((heads, Cc, hd), seed, hd ** -0.5)
        for h in range(heads):
            w[h * Cc:(h + 1) * Cc, h * hd:(h + 1) * hd] = blk[h]
        return ops.pad_rows(w.to(dt).cuda(), 128)

    w1, w2 = folded(1501), folded(1502)
    wc = ops.pad_rows(as_mode(rnd((heads, K), 1503), mode)[0], 128)

    def descs(outs, kwin):
        ds = (GemmDesc * 3)()
        for i, (w, n, o) in enumerate(((w1, N, outs[0]), (w2, N, outs[1]), (wc, heads, outs[2]))):
            d = ds[i]
            d.A, d.lda, d.W, d.ldw = a.data_ptr(), K, w.data_ptr(), K
            d.M, d.N, d.K, d.epi, d.ldc = M, n, K, EPI_DENSE, n
            if o.dtype == torch.float32 and mode != L4P_F32:
                d.out_f32 = o.data_ptr()
            else:
                d.out_T = o.data_ptr()
            if kwin and i < 2:
                d.kw_cols, d.kw_len = Cc, hd
        return ds

    def outs():
        return [torch.zeros(M, N, dtype=dt, device="cuda"), torch.zeros(M, N, dtype=dt, device="cuda"), torch.zeros(M, heads, dtype=torch.float32, device="cuda")]

    full, win, grp = outs(), outs(), outs()
    df, dw, dg = descs(full, False), descs(win, True), descs(grp, True)
    for i in range(3):
        _lib.check(lib.l4p_gemm(st, mode, C.byref(df[i])), "l4p_gemm")
        _lib.check(lib.l4p_gemm(st, mode, C.byref(dw[i])), "l4p_gemm(kw)")
    if mode != L4P_F32:
        _lib.check(lib.l4p_gemm_group(st, mode, dg, 3), "l4p_gemm_group(kw)")
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(full[i], win[i]), i
        if mode != L4P_F32:
            assert torch.equal(full[i], grp[i]), i
    ref = a.float().cpu() @ w1[:N].float().cpu().t()
    check(win[0], ref, mode, True)


def test_cu_masked_stream_runs_kernels(dev):
    """l4p_stream_create_cu_mask (plumbing of the sharded long-video path, parallel.cu_masked_stream): a stream confined to 32 CUs
    runs the engine's kernels with the same result as the default stream, and can be released."""
    import ctypes as C

    from l4p_amd import _lib

    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.l4p_stream_create_cu_mask(0, 32, C.byref(h)), "l4p_stream_create_cu_mask")
    assert h.value
    x = rnd((512, 1408), 901).cuda()
    g, b = rnd((1408,), 902).cuda(), rnd((1408,), 903).cuda()
    want = torch.empty(512, 1408, dtype=torch.bfloat16, device="cuda")
    got = torch.empty_like(want)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.l4p_layernorm(st, L4P_BF16, x.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-6, want.data_ptr(), None, 512, 1408), "l4p_layernorm")
    torch.cuda.synchronize()
    ext = torch.cuda.ExternalStream(h.value)
    _lib.check(lib.l4p_layernorm(h.value, L4P_BF16, x.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-6, got.data_ptr(), None, 512, 1408), "l4p_layernorm")
    ext.synchronize()
    assert torch.equal(got, want)
    del ext
    _lib.check(lib.l4p_stream_destroy(h.value), "l4p_stream_destroy")
    assert lib.l4p_stream_create_cu_mask(0, 0, C.byref(h)) != 0  # an empty CU range is refused
