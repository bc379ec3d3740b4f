"""Rows a6 / a7 of SURVEY.md §8 on the GPU.

* forward_windowed stitching (dense_heads.py:76-143) over THREE overlapping windows against the reference's committed
  golden samples (tests/golden/mini_T32_stitch.npz: depth with the inverse-depth LstSq seam aligner, flow with the
  skipped first frame of later windows, dyn_mask overwrite) — f32 engine 1e-3 relative-to-max, bf16 engine rel-L2.
* l4p_affine_align_solve / _apply (LstSqAffineAligner, aligner.py:29-66; safe_inverse, misc.py:48-62) as kernels:
  against torch.linalg.lstsq on the same data, including depths <= 0 (safe_inverse maps them to 0 on both sides of the
  solve and in the apply), an all-invalid overlap, n = 1 and the bit-reproducibility of the fixed-order reduction.
* LinearAligner(method="mean") (aligner.py:69-118) on the same kernels."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd.models.aligner import LinearAligner, LstSqAffineAligner
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch, sample_indices
from tests.test_encoder_dpt_gpu import build, rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DENSE = ["depth", "flow_2d_backward", "dyn_mask"]
KEYS = ["depth_est_b1thw", "flow_2d_backward_est_b2thw", "dyn_mask_est_b1thw"]


_ORACLE_T32 = {}


def _oracle_three_windows(sd, cfg, batch):
    """The CPU oracle's 3-window forward (30 s on the test box's host): the same for every engine precision, computed once."""
    if "ref" not in _ORACLE_T32:
        from oracle.l4p_oracle import OracleModel

        with torch.no_grad():
            _ORACLE_T32["ref"] = OracleModel(sd, cfg).forward(batch, DENSE)
    return _ORACLE_T32["ref"]


@pytest.mark.parametrize("precision", ["32-true", "bf16", "16-mixed"])
def test_three_window_stitch_vs_reference_goldens(dev, precision):
    cfg = ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    model = build(cfg, sd, precision)
    batch = make_batch(32, 12)
    gold = np.load(os.path.join(GOLD, "mini_T32_stitch.npz"))
    with torch.no_grad():
        out = model.forward({k: v.clone() for k, v in batch.items()}, DENSE)
    ref = _oracle_three_windows(sd, cfg, batch)
    torch.cuda.synchronize()
    drift = {}
    for key in KEYS:
        y = out[key].float().cpu()
        assert tuple(y.shape) == tuple(ref[key].shape) and y.shape[2] == 32, key
        s = y.reshape(-1)[sample_indices(y.numel())]
        g = torch.from_numpy(gold[key]).reshape(-1)
        if precision == "32-true":
            assert (s - g).abs().max() <= 1e-3 * g.abs().max(), (key, float((s - g).abs().max() / g.abs().max()))
            assert (y - ref[key]).abs().max() <= 1e-3 * ref[key].abs().max(), key
        else:
            drift[key] = rel_l2(y, ref[key])
            assert abs(rel_l2(s, g) - drift[key]) <= 0.1 * drift[key], (key, rel_l2(s, g), drift[key])  # samples ~ full tensor
    if drift:
        # the bf16 engine against the reference's OWN autocast drift over these 3 windows (tools/gen_golden_full_autocast.py)
        from tests.golden_utils import assert_bf16_within_reference_drift

        assert_bf16_within_reference_drift(drift, "mini_T32_stitch", precision=precision)


def _safe_inverse(x):
    y = torch.zeros_like(x)
    m = x > 0
    y[m] = 1.0 / x[m]
    return y


def _lstsq(pred, target, fn):
    a, b = fn(pred).reshape(pred.shape[0], -1, 1).double(), fn(target).reshape(pred.shape[0], -1, 1).double()
    A = torch.cat([a, torch.ones_like(a)], dim=-1)
    return torch.linalg.lstsq(A, b).solution[..., 0]  # [B, 2]


@pytest.mark.parametrize("pre_post", ["inverse", "identity"])
def test_affine_aligner_vs_lstsq_with_nonpositive_depths(dev, pre_post):
    g = torch.Generator().manual_seed(3)
    B, shape = 2, (1, 8, 56, 56)
    target = torch.rand((B,) + shape, generator=g) * 4 + 0.5
    pred = 1.0 / (0.7 * (1.0 / target) + 0.05) + 0.01 * torch.randn((B,) + shape, generator=g)
    # invalid / non-positive depths on both sides (safe_inverse -> 0), incl. exact zeros and negatives
    pred[0, 0, 0, :5] = 0.0
    pred[1, 0, 3, 10:14] = -1.5
    target[0, 0, 1, :3] = 0.0
    target[1, 0, 7, 50:] = -0.25
    fn = _safe_inverse if pre_post == "inverse" else (lambda x: x)
    al = LstSqAffineAligner(pre_post_fn=pre_post)
    al.solve(pred.cuda(), target.cuda())
    sol = al.sol.cpu().double()
    want = _lstsq(pred, target, fn)
    assert (sol - want).abs().max() <= 1e-5 * want.abs().max(), (sol, want)
    y = al.apply(pred.cuda()).cpu()
    ref = fn(want[:, 0].float().view(B, 1, 1, 1, 1) * fn(pred) + want[:, 1].float().view(B, 1, 1, 1, 1))
    assert (y - ref).abs().max() <= 1e-4 * ref.abs().max()
    if pre_post == "inverse":
        assert float(y[0, 0, 0, 0, 0]) == float(ref[0, 0, 0, 0, 0])  # a zero depth goes through scale*0 + shift, as in the reference
    # bit-reproducible: the same solve twice, and on a fresh aligner
    al2 = LstSqAffineAligner(pre_post_fn=pre_post)
    for _ in range(3):
        al2.solve(pred.cuda(), target.cuda())
        assert torch.equal(al2.sol, al.sol)


def test_affine_aligner_degenerate_overlaps(dev):
    # all-invalid overlap: every depth <= 0 -> both columns vanish; the normal equations are singular and the kernels return
    # (0, 0) (torch.linalg.lstsq's minimum-norm solution of the zero system), apply then yields safe_inverse(0) = 0
    al = LstSqAffineAligner(pre_post_fn="inverse")
    z = -torch.ones(1, 1, 8, 16, 16).cuda()
    al.solve(z, z)
    assert al.sol.cpu().tolist() == [[0.0, 0.0]]
    assert float(al.apply(z).abs().max()) == 0.0
    # a single sample: singular as well (one equation, two unknowns) -> (0, 0), no NaN
    one = torch.full((1, 1), 2.0).cuda()
    al.solve(one, one)
    assert bool(torch.isfinite(al.sol).all())
    # constant prediction (zero variance): singular -> finite
    c = torch.full((1, 64), 2.0).cuda()
    al.solve(c, torch.rand(1, 64).cuda() + 1)
    assert bool(torch.isfinite(al.sol).all())


def test_linear_aligner_mean_ratio(dev):
    g = torch.Generator().manual_seed(5)
    pred = torch.rand(2, 1, 8, 32, 32, generator=g) + 0.5
    target = pred * torch.tensor([1.7, 0.4]).view(2, 1, 1, 1, 1) * (1 + 0.01 * torch.randn(pred.shape, generator=g))
    for pre_post, fn in (("identity", lambda x: x), ("inverse", _safe_inverse)):
        al = LinearAligner(pre_post_fn=pre_post, method="mean")
        al.solve(pred.cuda(), target.cuda())
        ratios = (fn(target).reshape(2, -1) / (fn(pred).reshape(2, -1) + 1e-8)).double().mean(dim=1)
        assert (al.sol[:, 0].cpu().double() - ratios).abs().max() <= 1e-6 * ratios.abs().max()
        assert float(al.sol[:, 1].abs().max()) == 0.0
        y = al.apply(pred.cuda()).cpu()
        ref = fn(ratios.float().view(2, 1, 1, 1, 1) * fn(pred))
        assert (y - ref).abs().max() <= 1e-5 * ref.abs().max()
    with pytest.raises(ValueError):
        LinearAligner(method="mode")


@pytest.mark.parametrize("n", [8 * 224 * 224, 8 * 32 * 32, 7, 2, 1])
def test_linear_aligner_median_equals_torch_median(dev, n):
    """LinearAligner(method="median") (aligner.py:106-107): torch.median = the LOWER median of the float ratios, selected
    exactly (even and odd counts, duplicates at the selected rank, ratios of either sign in the identity mode, invalid depths
    in the inverse mode), against the oracle's restatement (== the imported reference, tools/gen_golden_joint.py)."""
    from oracle.l4p_oracle import linear_median_solve

    g = torch.Generator().manual_seed(n)
    pred = torch.rand(2, n, generator=g) + 0.5
    target = pred * torch.tensor([1.7, 0.4]).view(2, 1) * (1 + 0.05 * torch.randn(pred.shape, generator=g))
    if n > 100:
        target[0, : n // 3] = pred[0, : n // 3] * 1.7   # a third of the ratios nearly coincide around the median
        pred[1, 5:9] = 0.0                                # invalid depths: safe_inverse -> 0 -> ratio 0 / 1e-8
        target[0, 11:14] = -1.0
    for pre_post, tgt in (("inverse", target), ("identity", target - 1.2)):
        al = LinearAligner(pre_post_fn=pre_post, method="median")
        al.solve(pred.cuda(), tgt.cuda())
        want = linear_median_solve(pred, tgt, inverse=pre_post == "inverse")
        assert torch.equal(al.sol[:, 0].cpu(), want), (pre_post, al.sol[:, 0].cpu(), want)
        assert float(al.sol[:, 1].abs().max()) == 0.0
        y = al.apply(pred.cuda()).cpu()
        fn = _safe_inverse if pre_post == "inverse" else (lambda x: x)
        ref = fn(want.view(2, 1) * fn(pred))
        assert (y - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_quantile_and_rank_select_on_signed_values(dev):
    """l4p_quantile / l4p_select_rank accept finite floats of either sign (order-preserving key), incl. -0.0 next to +0.0."""
    import ctypes as C

    from l4p_amd import _lib

    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(2052, dtype=torch.int32, device="cuda")
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(10007, generator=g)
    x[:50] = 0.0
    x[50:80] = -0.0
    xd = x.cuda()
    for q in (0.0, 0.02, 0.5, 0.77, 1.0):
        _lib.check(lib.l4p_quantile(st, xd.data_ptr(), x.numel(), q, ws.data_ptr(), out.data_ptr()), "l4p_quantile")
        want = float(torch.quantile(x, q))
        assert abs(float(out.cpu()) - want) <= 1e-6 * max(abs(want), 1e-3), (q, float(out.cpu()), want)
    srt = torch.sort(x).values
    for r in (0, 1, 4999, 5003, 10006):
        _lib.check(lib.l4p_select_rank(st, xd.data_ptr(), x.numel(), r, ws.data_ptr(), out.data_ptr()), "l4p_select_rank")
        assert float(out.cpu()) == float(srt[r]), (r, float(out.cpu()), float(srt[r]))


def test_depth_stitch_with_linear_aligner(dev):
    """VideoMAEDepthDPTHead(align_type="linear") (dense_heads.py:146-170): a 2-window depth stitch with the scale-only
    LinearAligner(method="mean") against the oracle's restatement of aligner.py:91-118, and it must differ from the affine one."""
    from l4p_amd.models.aligner import LinearAligner
    from oracle.l4p_oracle import OracleModel

    cfg = ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    model = build(cfg, sd, "32-true")
    batch = make_batch(24, 2)
    with torch.no_grad():
        affine = model.forward({k: v.clone() for k, v in batch.items()}, ["depth"])["depth_est_b1thw"].float().cpu()
        model.l4p_model.task_heads["depth"].overlap_aligner_type = LinearAligner
        y = model.forward({k: v.clone() for k, v in batch.items()}, ["depth"])["depth_est_b1thw"].float().cpu()
        om = OracleModel(sd, cfg)
        om.depth_align_type = "linear"
        ref = om.forward(batch, ["depth"])["depth_est_b1thw"]
    torch.cuda.synchronize()
    assert (y - ref).abs().max() <= 1e-3 * ref.abs().max(), float((y - ref).abs().max() / ref.abs().max())
    assert (y - affine).abs().max() > 1e-4 * ref.abs().max()  # a different aligner: later windows differ
    assert torch.equal(y[:, :, :8], affine[:, :, :8])          # the first window is never re-aligned
