"""Parity of the two-workgroups-per-CU bf16 GEMM (csrc/gemm4w.hpp: 4-wave workgroups, 256 x 128 tiles, LDS-DMA through buffer
descriptors, 64-deep k-tiles, A double- and W single-buffered in 80 KB of LDS) against plain PyTorch fp32 on the same bf16-rounded inputs, at the shapes the bench runs on it
and at the edges of its pipeline (1, 2, 3, 5.5, 18.4 k-tiles: K not a multiple of 64 reads its tail through the descriptor's range
check; ragged M / N; row maps; every epilogue family it can be handed).
The tests are tests/test_gemm8p_gpu.py's own bodies with the launcher forced onto this form (l4p_set_knob("gemm_4w", 2)) and the profiler tag
asserted to be " 4w ".  Reference call sites: modeling_finetune.py:169-190,62-69, sam/transformer.py:223-245,
mask_decoder.py:58-66,136-139."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd import ops
from tests import test_gemm8p_gpu as t8
from tests.test_kernels_gpu import as_mode, check, rnd


@pytest.fixture(autouse=True)
def _force_4w(monkeypatch, probe_kernels, knob):
    knob("gemm_4w", 2)
    monkeypatch.setattr(t8, "FORM_TAG", " 4w ")


@pytest.mark.parametrize("M,N,K", [(8192, 6144, 1408),   # fc1: 1536 tiles = three rounds of the chip's 512 slots
                                     (8192, 4608, 1408),   # QKV's shape
                                     (8192, 1408, 1408),   # proj: 352 tiles, part of one round
                                     (16384, 2048, 704),   # 22 k-steps
                                     (8200, 1416, 1216),   # ragged M and N (tile tails), 38 k-steps
                                     (16384, 1408, 1176),  # K % 64 = 24: three valid chunks in the last k-tile
                                     (8192, 2048, 352),    # 5.5 k-tiles (the mask product's K)
                                     (8192, 2048, 32),     # half a k-tile: prologue only, tail in the prologue
                                     (8192, 2048, 64),     # one
                                     (8192, 2048, 128),    # two: both A buffers once
                                     (8192, 2048, 200)])   # 3.125: the first A buffer reused, one valid chunk in the tail
def test_gemm4w_dense_bias(dev, M, N, K):
    t8.test_gemm8p_dense_bias(dev, M, N, K)


def test_gemm4w_gelu_and_f32_residual_inplace(dev):
    t8.test_gemm8p_gelu_and_f32_residual_inplace(dev)


def test_gemm4w_two_residuals_T(dev):
    t8.test_gemm8p_two_residuals_T(dev)


def test_gemm4w_tracker_i2t_out_res_inplace_and_row_maps(dev):
    t8.test_gemm8p_tracker_i2t_out_res_inplace_and_row_maps(dev)


def test_gemm4w_qkv_epilogue_and_batch4_attention(dev):
    t8.test_gemm8p_qkv_epilogue_and_batch4_attention(dev)


def test_gemm4w_conv_transpose_upscaling(dev):
    t8.test_gemm8p_conv_transpose_upscaling(dev)


def test_gemm4w_maskdot_large(dev, knob):
    t8.test_gemm8p_maskdot_large(dev, knob)


def test_gemm4w_equals_8p_bitwise_and_is_deterministic(dev, monkeypatch, knob):
    """Same k order inside a tile (ascending k, one accumulator per output) on both forms: the float outputs agree bit for bit,
    and two launches of the 4w form agree with each other (no inter-workgroup dependence)."""
    M, N, K = 8192, 4608, 1408
    a, _ = as_mode(rnd((M, K), 300), t8.MODE)
    w, _ = as_mode(rnd((N, K), 301, K ** -0.5), t8.MODE)
    wp = ops.pad_rows(w, 256)
    bias = rnd((N,), 302).cuda()
    with t8.prof_tags() as p:
        _, y1 = ops.gemm(a, wp, N, bias=bias, out_f32=True, out_T=True)
        _, y2 = ops.gemm(a, wp, N, bias=bias, out_f32=True, out_T=True)
    p.assert_8p()
    assert sum(int(ln[2]) for ln in p.lines if ln[0] == "gemm") == 2, p.lines
    knob("gemm_4w", 0)
    monkeypatch.setattr(t8, "FORM_TAG", " 8p ")
    with t8.prof_tags() as p:
        _, y8 = ops.gemm(a, wp, N, bias=bias, out_f32=True, out_T=True)
    p.assert_8p()
    assert torch.equal(y1, y2)
    assert torch.equal(y1, y8)
