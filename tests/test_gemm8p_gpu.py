"""Direct parity of the 8-phase 256x256 bf16 kernel (csrc/gemm8p.hpp) — the kernel that carries the batch-4 encoder
linears, the tracker's tall GEMMs and the N = 256 DPT convs of the benchmarked workload — against plain PyTorch fp32 on
the same bf16-rounded inputs, at the SHAPES THE BENCH RUNS.  tests/test_kernels_gpu.py's shapes all fall on the 128x128
kernel (the 8-phase kernel is selected for >= 192..256 tiles of 256x256, gemm_launch.inc), so every test here asserts
through the event profiler's tag (l4p_prof_detail) that the launch really was the 8-phase kernel.

Tolerances are tests/test_kernels_gpu.py's: float outputs 1e-3 * max|ref|; bf16 outputs rel-L2 <= 3e-3 and max error
<= 1 bf16 ulp of max|ref|.  Reference call sites: modeling_finetune.py:169-190 (qkv / proj), :62-69 (fc1 / fc2),
sam/transformer.py:223-245 (i2t out projection, K = 704), mask_decoder.py:58-66,136-139 (up-scaling ConvTranspose, mask
product), dpt_block.py:110-157 (3x3x3 convs)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from l4p_amd import _lib, ops
from l4p_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, EPI_DENSE, EPI_MASKDOT, L4P_BF16, L4P_F16, GemmDesc
from tests.test_kernels_gpu import as_mode, check, rnd

MODE = L4P_BF16  # (set per test by the fixture below: every test of this file runs on both 16-bit engine types)


@pytest.fixture(autouse=True, params=[L4P_BF16, L4P_F16], ids=["bf16", "f16"])
def _engine_mode(request, monkeypatch):
    import sys

    monkeypatch.setattr(sys.modules[__name__], "MODE", request.param)


FORM_TAG = " 8p "  # the kernel form every test here asserts (tests/test_gemm4w_gpu.py re-runs some of them on the " 4w " form)


class prof_tags:
    """Collect the (class, tag) lines of every launch inside the block (l4p_prof_detail)."""

    def __enter__(self):
        self.lib = _lib.load()
        torch.cuda.synchronize()
        self.lib.l4p_prof_reset()
        self.lib.l4p_prof_enable(1)
        self.lines = []
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        self.lib.l4p_prof_enable(0)
        n = self.lib.l4p_prof_detail(None, 0)
        buf = C.create_string_buffer(int(n) + 16)
        self.lib.l4p_prof_detail(buf, len(buf))
        self.lines = [ln.split("\t") for ln in buf.value.decode().splitlines() if ln]
        self.lib.l4p_prof_reset()
        return False

    def assert_8p(self, cls="gemm", n=1):
        tags = [ln[1] for ln in self.lines if ln[0] == cls]
        assert len(tags) >= n and all(FORM_TAG in t for t in tags), f"expected the{FORM_TAG}kernel, launches were: {self.lines}"


@pytest.mark.parametrize("M,N,K", [(8192, 6144, 1408),   # fc1 (+ GELU below)
                                     (8192, 1408, 6144),   # fc2: 192 tiles
                                     (8192, 1408, 1408),   # proj
                                     (16384, 2048, 704),   # odd number of k-tiles (11)
                                     (8200, 1416, 1216),   # ragged M and N (tile tails), 19 k-tiles
                                     (16384, 1408, 1176)]) # K not a multiple of the k-tile (18.4 tiles; K % 8 == 0)
def test_gemm8p_dense_bias(dev, M, N, K):
    a, a_ref = as_mode(rnd((M, K), 10), MODE)
    w, w_ref = as_mode(rnd((N, K), 11, K ** -0.5), MODE)
    bias = rnd((N,), 12)
    ref = a_ref @ w_ref.t() + bias
    with prof_tags() as p:
        yT, yf = ops.gemm(a, ops.pad_rows(w, 256), N, bias=bias.cuda(), out_f32=True, out_T=True)
    p.assert_8p()
    check(yf, ref, MODE, False)
    check(yT, ref, MODE, True)


def test_gemm8p_gelu_and_f32_residual_inplace(dev):
    """fc1's GELU epilogue, and fc2's form: float residual stream read and written IN PLACE (out_f32 aliases res1)."""
    M, N, K = 8192, 6144, 1408
    a, a_ref = as_mode(rnd((M, K), 20), MODE)
    w, w_ref = as_mode(rnd((N, K), 21, K ** -0.5), MODE)
    bias = rnd((N,), 22)
    with prof_tags() as p:
        h, _ = ops.gemm(a, ops.pad_rows(w, 256), N, bias=bias.cuda(), act=ACT_GELU)
    p.assert_8p()
    href = F.gelu(a_ref @ w_ref.t() + bias)
    check(h, href, MODE, True)
    # fc2 on the kernel's own (bf16) hidden activations, residual in place
    w2, w2_ref = as_mode(rnd((K, N), 23, N ** -0.5), MODE)
    b2 = rnd((K,), 24)
    res = rnd((M, K), 25, 3.0)
    x = res.clone().cuda()
    with prof_tags() as p:
        ops.gemm(h, ops.pad_rows(w2, 256), K, bias=b2.cuda(), res1=x, out=x)
    p.assert_8p()
    check(x, h.float().cpu() @ w2_ref.t() + b2 + res, MODE, False)


@pytest.mark.parametrize("M,N,K,sk", [(2048, 1408, 6144, 5),   # the batch-1 MLP-out projection: 48 tiles x 5 slices, 19.2 k-tiles each
                                        (2048, 1408, 6144, 4),   # as the encoder runs it: 64 tiles of 256x192 x 4 slices = 256 workgroups
                                        (2304, 1416, 4104, 4)])  # ragged M / N / K (64.1 k-tiles: slices of 16, 16, 16, 17)
def test_gemm8p_splitk_f32_residual_inplace(dev, M, N, K, sk):
    """Split-K on the 8-phase kernel: slices of the 256x256 tiles leave float partials, splitk_finish_kernel sums them in
    slice order and applies bias + the float residual in place (the encoder's fc2 at batch 1, api.hip:enc_fc2_splitk)."""
    a, a_ref = as_mode(rnd((M, K), 26), MODE)
    w, w_ref = as_mode(rnd((N, K), 27, K ** -0.5), MODE)
    bias = rnd((N,), 28)
    res = rnd((M, N), 29, 3.0)
    x = res.clone().cuda()
    wp = ops.pad_rows(w, 256)
    partial = torch.empty((sk, M, N), dtype=torch.float32, device="cuda")
    d = GemmDesc()
    d.A, d.lda, d.W, d.ldw = a.data_ptr(), K, wp.data_ptr(), K
    d.M, d.N, d.K = M, N, K
    d.bias = bias.cuda().data_ptr()
    bias_dev = bias.cuda()
    d.bias = bias_dev.data_ptr()
    d.res1, d.res_f32, d.ldr = x.data_ptr(), 1, N
    d.out_f32, d.ldc = x.data_ptr(), N
    d.epi = EPI_DENSE
    d.splitk, d.partial = sk, partial.data_ptr()
    with prof_tags() as p:
        _lib.check(_lib.load().l4p_gemm(torch.cuda.current_stream().cuda_stream, MODE, C.byref(d)), "l4p_gemm")
    tags = [ln[1] for ln in p.lines if ln[0] == "gemm"]
    assert tags and all(f" 8p sk{sk} " in t for t in tags), tags
    if (M, N, K, sk) == (2048, 1408, 6144, 4):
        assert all("t256x192" in t for t in tags), tags  # (gemm_launch.inc: the split-K form of the 4 x 6 wave tile)
    check(x, a_ref @ w_ref.t() + bias + res, MODE, False)
    # run-to-run bit-reproducible (fixed summation order of the slices)
    x2 = res.clone().cuda()
    d.res1 = d.out_f32 = x2.data_ptr()
    _lib.check(_lib.load().l4p_gemm(torch.cuda.current_stream().cuda_stream, MODE, C.byref(d)), "l4p_gemm")
    torch.cuda.synchronize()
    assert torch.equal(x, x2)


def test_gemm8p_two_residuals_T(dev):
    """DPT skip connections: two residuals stored in the engine dtype + ReLU."""
    M, N, K = 65536, 256, 256
    a, a_ref = as_mode(rnd((M, K), 30), MODE)
    w, w_ref = as_mode(rnd((N, K), 31, K ** -0.5), MODE)
    bias = rnd((N,), 32)
    r1, r1_ref = as_mode(rnd((M, N), 33), MODE)
    r2, r2_ref = as_mode(rnd((M, N), 34), MODE)
    with prof_tags() as p:
        y, _ = ops.gemm(a, ops.pad_rows(w, 256), N, bias=bias.cuda(), act=ACT_RELU, res1=r1, res2=r2)
    p.assert_8p()
    check(y, F.relu(a_ref @ w_ref.t() + bias) + r1_ref + r2_ref, MODE, True)


def test_gemm8p_tracker_i2t_out_res_inplace_and_row_maps(dev):
    """The tracker's image->token output projection (M = N_q * 2048 key rows, N = 1408, K = 704) with its f32 residual
    updated in place; then the same GEMM restricted by the row maps to the second temporal half of every query's token
    block (sparse_heads.py:406-448: rows (m / 1024) * 2048 + 1024 + m % 1024 of A and of the output)."""
    Nq, P, Cc, K = 64, 2048, 1408, 704
    M = Nq * P
    a, a_ref = as_mode(rnd((M, K), 40), MODE)
    w, w_ref = as_mode(rnd((Cc, K), 41, K ** -0.5), MODE)
    bias = rnd((Cc,), 42)
    keys = rnd((M, Cc), 43)
    x = keys.clone().cuda()
    with prof_tags() as p:
        ops.gemm(a, ops.pad_rows(w, 256), Cc, bias=bias.cuda(), res1=x, out=x)
    p.assert_8p()
    ref = a_ref @ w_ref.t() + bias + keys
    check(x, ref, MODE, False)
    del x
    # row-mapped: logical rows = second half (1024 rows) of each of the Nq blocks of 2048
    half = P // 2
    out = torch.zeros((M, Cc), dtype=torch.float32, device="cuda")
    d = GemmDesc()
    wp = ops.pad_rows(w, 256)
    bc = bias.cuda()
    d.A, d.lda, d.W, d.ldw = a.data_ptr(), K, wp.data_ptr(), K
    d.M, d.N, d.K = Nq * half, Cc, K
    d.bias = bc.data_ptr()
    d.out_f32, d.ldc = out.data_ptr(), Cc
    d.epi = EPI_DENSE
    d.a_gr, d.a_gs, d.a_go = half, P, half
    d.c_gr, d.c_gs, d.c_go = half, P, half
    with prof_tags() as p:
        _lib.check(_lib.load().l4p_gemm(torch.cuda.current_stream().cuda_stream, MODE, C.byref(d)), "l4p_gemm(row map)")
    p.assert_8p()
    o = out.cpu().view(Nq, P, Cc)
    assert float(o[:, :half].abs().max()) == 0.0, "rows outside the map were written"
    check(o[:, half:], (ref - keys).view(Nq, P, Cc)[:, half:], MODE, False)


def test_gemm8p_qkv_epilogue_and_batch4_attention(dev):
    """The benchmarked encoder shapes: EPI_QKV at M = 8192, N = 4608 (8-phase kernel: q dense, K in tile order, V
    transposed) feeding the un-split (SPLIT = 1) hand-scheduled attention kernel that only batch >= 4 selects."""
    B, S, H, Dh = 4, 2048, 16, 88
    Cc = H * Dh
    x, x_ref = as_mode(rnd((B * S, Cc), 50), MODE)
    wqkv = rnd((3 * Cc, Cc), 51, Cc ** -0.5)
    qb, vb = rnd((Cc,), 52) * 0.1, rnd((Cc,), 53) * 0.1
    wp = torch.zeros(3, H, ops.DP, Cc)
    wp[:, :, :Dh] = wqkv.view(3, H, Dh, Cc)
    bp = torch.zeros(3, H, ops.DP)
    bp[0, :, :Dh] = qb.view(H, Dh)
    bp[2, :, :Dh] = vb.view(H, Dh)
    w, w_ref = as_mode(wp.view(3 * H * ops.DP, Cc), MODE)
    with prof_tags() as p:
        qs = Dh ** -0.5 * 1.4426950408889634  # the engine's form: q leaves the projection in the exp2 domain
        q, kt, vt = ops.qkv_gemm(x, ops.pad_rows(w, 256), bp.view(-1).cuda(), B, S, H, q_scale=qs)
    p.assert_8p()
    full = (x_ref @ w_ref.t() + bp.view(-1)).view(B, S, 3, H, ops.DP)
    check(q.view(B, S, H, ops.DP), full[:, :, 0] * qs, MODE, True)
    check(vt, full[:, :, 2].permute(0, 2, 3, 1), MODE, True)
    check(kt, ops.k_tile_order(full[:, :, 1].contiguous().to(ops.torch_dtype(MODE))).float(), MODE, True)
    q_ref = q.float().cpu().view(B, S, H, ops.DP).permute(0, 2, 1, 3)
    k_ref = full[:, :, 1].to(ops.torch_dtype(MODE)).float().permute(0, 2, 1, 3)
    v_ref = vt.float().cpu().permute(0, 1, 3, 2)
    out = ops.attention(q, kt, vt, Dh, scale=0.0)
    for b in range(B):  # one clip at a time: the score matrices of a clip are 16 x 2048 x 2048 floats
        attn = torch.softmax((q_ref[b] * 0.6931471805599453) @ k_ref[b].transpose(-2, -1), dim=-1)
        ref = (attn @ v_ref[b])[..., :Dh].transpose(0, 1).reshape(S, Cc)
        check(out[b * S:(b + 1) * S], ref, MODE, True)


def test_gemm8p_conv_transpose_upscaling(dev):
    """The tracker's first up-scaling ConvTranspose3d(1408 -> 352, k = s = 2) for 8 queries: M = 16384, N = 2816, K = 1408,
    scatter epilogue (EPI_CONVT)."""
    Nq, T, h, w_, Cin, cout, k = 8, 8, 16, 16, 1408, 352, (2, 2, 2)
    x, x_ref = as_mode(rnd((Nq, T, h, w_, Cin), 60), MODE)
    wt = rnd((Cin, cout) + k, 61, Cin ** -0.5)
    bias = rnd((cout,), 62)
    wT, w_ref = as_mode(wt.permute(2, 3, 4, 1, 0).reshape(8 * cout, Cin), MODE)
    w5 = w_ref.view(2, 2, 2, cout, Cin).permute(4, 3, 0, 1, 2)
    ref = F.conv_transpose3d(x_ref.permute(0, 4, 1, 2, 3), w5, bias, stride=k).permute(0, 2, 3, 4, 1)
    with prof_tags() as p:
        y = ops.conv_transpose(x, ops.pad_rows(wT, 256), cout, k, bias_taps=bias.repeat(8).cuda())
    p.assert_8p()
    check(y, ref, MODE, True)


def test_gemm8p_maskdot_large(dev, knob):
    """L4P_EPI_MASKDOT on the 8-phase kernel (M = 65536 >= the 256-tile threshold): GELU(ConvTranspose(1,2,2)) contracted
    with the per-query hyper-network vectors (mask_decoder.py:136-139), against the unfused statement.  Two forms: the
    matrix-pipe contraction (default; the activated row and the hyper vectors are rounded to the engine type first, as the
    reference's autocast holds them) against the statement on operands rounded the same way, and the all-VALU form
    (knob maskdot_mfma = 0: float row, float dot products) against the unrounded statement."""
    Nq, T, h, w_, Cin, d1 = 4, 16, 32, 32, 352, 176
    x, x_ref = as_mode(rnd((Nq, T, h, w_, Cin), 90), MODE)
    wt = rnd((Cin, d1, 1, 2, 2), 91, Cin ** -0.5)
    bias = rnd((d1,), 92)
    hyper = rnd((Nq, 3, d1), 93)
    wm = wt.permute(2, 3, 4, 1, 0).reshape(4, d1, Cin)
    wp, w_ref = as_mode(wm.reshape(4 * d1, Cin), MODE)
    up = F.gelu(F.conv_transpose3d(x_ref.permute(0, 4, 1, 2, 3), w_ref.view(1, 2, 2, d1, Cin).permute(4, 3, 0, 1, 2), bias,
                                   stride=(1, 2, 2)))
    ref = torch.einsum("nic,nctyx->nityx", hyper, up)
    M = Nq * T * h * w_
    cpt = d1 // 32  # 176 = 5.5 chunks: d1 must be padded to a multiple of 32 per tap
    d1p = (d1 + 31) // 32 * 32
    cpt = d1p // 32
    wpp, _ = as_mode(F.pad(wm, (0, 0, 0, d1p - d1)).reshape(4 * d1p, Cin), MODE)
    hp = torch.zeros(Nq, 3, d1p)
    hp[..., :d1] = hyper
    hp = hp.cuda()
    bp = F.pad(bias, (0, d1p - d1)).repeat(4).cuda()
    partial = torch.empty(4 * cpt, 3, M, dtype=torch.float32, device="cuda")
    wpad = ops.pad_rows(wpp, 256)
    d = GemmDesc()
    d.A, d.lda, d.W, d.ldw = x.data_ptr(), Cin, wpad.data_ptr(), Cin
    d.M, d.N, d.K = M, 4 * d1p, Cin
    d.bias, d.act = bp.data_ptr(), ACT_GELU
    d.out_f32 = partial.data_ptr()
    d.epi, d.Cout = EPI_MASKDOT, d1p
    d.hyper, d.hyper_rows = hp.data_ptr(), M // Nq
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    masks = torch.empty(Nq, 3, T, 2 * h, 2 * w_, dtype=torch.float32, device="cuda")
    td = ops.torch_dtype(MODE)
    ref_rounded = torch.einsum("nic,nctyx->nityx", hyper.to(td).float(), up.to(td).float())
    for mfma, want in ((1, ref_rounded), (0, ref)):
        knob("maskdot_mfma", mfma)
        partial.fill_(float("nan"))
        with prof_tags() as p:
            _lib.check(lib.l4p_gemm(st, MODE, C.byref(d)), "l4p_gemm(maskdot)")
        p.assert_8p()
        _lib.check(lib.l4p_mask_gather(st, partial.data_ptr(), masks.data_ptr(), Nq, T, h, w_, cpt), "l4p_mask_gather")
        torch.cuda.synchronize()
        if mfma:
            # (the kernel rounds ITS float GELU values to the engine type, the statement rounds torch's: values next to a rounding
            #  boundary land on either side - an ulp of one term in a 176-term sum)
            err = float((masks.cpu() - want).abs().max() / want.abs().max())
            assert err <= (3e-3 if td == torch.bfloat16 else 4e-4), err
        else:
            check(masks, want, MODE, False)
        # (both forms are the same function up to the operand rounding: a 176-term dot product of values rounded to 8 / 11 bits)
        rel = float((masks.cpu() - ref).norm() / ref.norm())
        assert rel <= (6e-3 if td == torch.bfloat16 else 8e-4), (mfma, rel)


def _conv_rows_reference(x_ref, w_ref, bias, rows, stride=(1, 1, 1)):
    """fp32 reference of SELECTED output voxels of a 3x3x3 / pad 1 conv (channels-last): rows = flat output indices.
    (The full M = 262144, K = 6912 problem is 0.93 TFLOP — minutes on the host; whole tiles + scattered rows are not.)"""
    B, Ti, Hi, Wi, Cin = x_ref.shape
    st, sh, sw = stride
    To, Ho, Wo = (Ti - 1) // st + 1, (Hi - 1) // sh + 1, (Wi - 1) // sw + 1
    xp = F.pad(x_ref, (0, 0, 1, 1, 1, 1, 1, 1))  # pad W, H, T by one voxel
    wo = rows % Wo
    r = rows // Wo
    ho = r % Ho
    r = r // Ho
    to = r % To
    b = r // To
    cols = []
    for dt in range(3):
        for dh in range(3):
            for dw in range(3):
                cols.append(xp[b, to * st + dt, ho * sh + dh, wo * sw + dw])  # [R, Cin]
    a = torch.cat(cols, dim=1)  # [R, 27*Cin], k = tap*Cin + c
    return a @ w_ref.t() + bias


@pytest.mark.parametrize("shape,cout", [((4, 16, 64, 64, 256), 256),   # refinenet RCU convs of a B = 4 step: K = 6912
                                         ((4, 16, 32, 32, 512), 256),   # layer_rn: K = 13824
                                         ((2, 16, 56, 56, 256), 256)])  # plane 3136 = 12.25 tiles: walk falls back, ragged rows
def test_gemm8p_conv3d(dev, shape, cout, knob):
    knob("conv_halo", 0)  # the implicit-GEMM form (the LDS-halo kernel has its own tests below)
    B, T, H, W, Cin = shape
    x, x_ref = as_mode(rnd(shape, 70), MODE)
    w = rnd((cout, Cin, 3, 3, 3), 71, (27 * Cin) ** -0.5)
    bias = rnd((cout,), 72)
    wT, w_ref = as_mode(w.permute(0, 2, 3, 4, 1).reshape(cout, 27 * Cin), MODE)
    M = B * T * H * W
    skip, s_ref = as_mode(rnd((B, T, H, W, cout), 73), MODE)
    with prof_tags() as p:
        y, yr = ops.conv3d_k3(x, ops.pad_rows(wT, 256), cout, bias=bias.cuda(), act=ACT_NONE, res1=skip, relu_copy=True)
    p.assert_8p("conv3d")
    g = torch.Generator().manual_seed(7)
    rows = torch.cat([torch.arange(0, 512),                       # first two tiles (t = 0 border, h = 0 border)
                      torch.arange(M - 512, M),                   # last tiles (far borders)
                      torch.arange(M // 2 - 256, M // 2 + 256),   # a tile pair in the interior / across a plane seam
                      torch.randint(0, M, (3072,), generator=g)])
    ref = _conv_rows_reference(x_ref, w_ref, bias, rows) + s_ref.reshape(M, cout)[rows]
    got = y.reshape(M, cout)[rows.cuda()]
    check(got, ref, MODE, True)
    check(yr.reshape(M, cout)[rows.cuda()], F.relu(ref), MODE, True)
    # every row at least finite and of the right scale (catches a tile that was never written)
    assert bool(torch.isfinite(y.float()).all())
    rms = y.float().pow(2).mean(dim=-1).sqrt().reshape(-1)
    assert float(rms.min()) > 0.05 * float(rms.mean())


def _assert_halo(p, n=1):
    tags = [ln[1] for ln in p.lines if ln[0] == "conv3d"]
    assert len(tags) >= n and all(" halo " in t for t in tags), f"expected the LDS-halo conv kernel, launches were: {p.lines}"


def _halo_rows(B, T, H, W, th):
    """Output voxels that exercise every kind of position of the LDS-halo kernel's 2 x th x 16 blocks: whole first / last
    blocks (volume borders in t, h, w: zero halo rows), block seams inside the volume, batch seams, scattered rows."""
    g = torch.Generator().manual_seed(7)
    M = B * T * H * W
    idx = torch.arange(M).reshape(B, T, H, W)
    picks = [idx[0, :2, :th, :16], idx[-1, -2:, -th:, -16:],             # first and last block
             idx[0, 1:3, th - 1:th + 1, 14:18],                          # a corner where eight blocks meet
             idx[B // 2, T // 2 - 1:T // 2 + 1, H // 2 - 1:H // 2 + 1, :],  # two full image rows across w blocks
             idx[:, 0, 0, 0], idx[:, -1, -1, -1]]                        # batch seams
    rows = torch.cat([p.reshape(-1) for p in picks] + [torch.randint(0, M, (3072,), generator=g)])
    return rows


@pytest.mark.parametrize("shape,cout,act,res", [
    ((4, 16, 64, 64, 256), 256, ACT_NONE, 2),    # refinenet RCU conv2 of a B = 4 step: two T residuals + relu copy
    ((4, 16, 64, 64, 256), 256, ACT_RELU, 0),    # RCU conv1
    ((4, 16, 32, 32, 512), 256, ACT_NONE, 0),    # layer_rn: 16 channel slices
    ((1, 16, 128, 128, 256), 128, ACT_NONE, 0),  # head1 at batch 1: 512-voxel blocks (2 x 16 x 16), N = 128
    ((2, 4, 224, 224, 128), 128, ACT_RELU, 0),   # head2 geometry (224 = 14 blocks), short in t: every block touches a t border
])
def test_conv3_halo(dev, shape, cout, act, res):
    """csrc/conv3_halo.hpp (input block staged once per 32-channel slice in LDS, 27 taps walked out of LDS) against fp32
    torch on the same bf16-rounded inputs (dpt_block.py:110-157,406-414), at the shapes the benchmark runs."""
    B, T, H, W, Cin = shape
    x, x_ref = as_mode(rnd(shape, 80), MODE)
    w = rnd((cout, Cin, 3, 3, 3), 81, (27 * Cin) ** -0.5)
    bias = rnd((cout,), 82)
    wT, w_ref = as_mode(w.permute(0, 2, 3, 4, 1).reshape(cout, 27 * Cin), MODE)
    M = B * T * H * W
    kw = {}
    if res:
        s1, s1_ref = as_mode(rnd((B, T, H, W, cout), 83), MODE)
        s2, s2_ref = as_mode(rnd((B, T, H, W, cout), 84), MODE)
        kw = dict(res1=s1, res2=s2, relu_copy=True)
    with prof_tags() as p:
        out = ops.conv3d_k3(x, ops.pad_rows(wT, 256), cout, bias=bias.cuda(), act=act, **kw)
    _assert_halo(p)
    y, yr = out if res else (out, None)
    rows = _halo_rows(B, T, H, W, 8 if cout == 256 else 16)
    ref = _conv_rows_reference(x_ref, w_ref, bias, rows)
    if act == ACT_RELU:
        ref = F.relu(ref)
    if res:
        ref = ref + s1_ref.reshape(M, cout)[rows] + s2_ref.reshape(M, cout)[rows]
    check(y.reshape(M, cout)[rows.cuda()], ref, MODE, True)
    if res:
        check(yr.reshape(M, cout)[rows.cuda()], F.relu(ref), MODE, True)
    assert bool(torch.isfinite(y.float()).all())
    rms = y.float().pow(2).mean(dim=-1).sqrt().reshape(-1)
    assert float(rms.min()) > 0.05 * float(rms.mean())  # (a block that was never written, or written at the wrong voxels)


def test_conv3_halo_equals_implicit_gemm_form(dev, knob):
    """Same arithmetic, different data movement: on the same inputs the LDS-halo kernel and the implicit-GEMM kernel agree to
    the summation-order level (the k order differs: channel slice outermost vs tap outermost) on EVERY output voxel."""
    shape, cout = (2, 8, 64, 64, 256), 256
    x, _ = as_mode(rnd(shape, 90), MODE)
    w = rnd((cout, 256, 3, 3, 3), 91, (27 * 256) ** -0.5)
    wT, _ = as_mode(w.permute(0, 2, 3, 4, 1).reshape(cout, 27 * 256), MODE)
    wp = ops.pad_rows(wT, 256)
    bias = rnd((cout,), 92).cuda()
    with prof_tags() as p:
        a = ops.conv3d_k3(x, wp, cout, bias=bias)
    _assert_halo(p)
    knob("conv_halo", 0)
    with prof_tags() as p:
        b = ops.conv3d_k3(x, wp, cout, bias=bias)
    assert all(" halo " not in ln[1] for ln in p.lines)
    d = (a.float() - b.float()).abs()
    assert float(d.max()) <= 2 ** -6 * float(b.float().abs().max()), float(d.max())  # <= 2 bf16 ulp of the maximum, everywhere
    assert float((a.float() - b.float()).norm() / b.float().norm()) <= 3e-3


@pytest.mark.parametrize("B,T,lo,hi", [(1, 16, (128, 128), (224, 224)), (2, 4, (60, 72), (112, 128))])
def test_conv3_halo_fused_upsample_equals_upsample_then_conv(dev, probe_kernels, B, T, lo, hi):
    """l4p_gemm_desc.ups_hi / ups_wi: the bilinear (align_corners) up-sampling in front of a 3x3x3 conv formed inside the LDS-halo
    kernel's loader (dpt_head.py:79-84: interpolate -> head conv) - equal, bit for bit, to l4p_upsample_trilinear followed by the
    conv on the stored volume: the loader reproduces that kernel's arithmetic and rounding.  Full-size head shape and a small,
    non-square one with a fractional scale in both axes."""
    Cin, cout = 128, 128
    x, _ = as_mode(rnd((B, T, lo[0], lo[1], Cin), 400), MODE)
    w = rnd((cout, Cin, 3, 3, 3), 401, (27 * Cin) ** -0.5)
    wT, _ = as_mode(w.permute(0, 2, 3, 4, 1).reshape(cout, 27 * Cin), MODE)
    wp = ops.pad_rows(wT, 256)
    bias = rnd((cout,), 402).cuda()
    up = ops.upsample_trilinear(x, (T, hi[0], hi[1]), align_corners=True)
    with prof_tags() as p:
        want = ops.conv3d_k3(up, wp, cout, bias=bias, act=ACT_RELU)
    assert any(" halo " in ln[1] for ln in p.lines), p.lines
    with prof_tags() as p:
        got = ops.conv3d_k3(x, wp, cout, bias=bias, act=ACT_RELU, ups_to=hi)
    assert any(" halo ups " in ln[1] for ln in p.lines), p.lines
    torch.cuda.synchronize()
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    again = ops.conv3d_k3(x, wp, cout, bias=bias, act=ACT_RELU, ups_to=hi)
    assert torch.equal(got, again)
