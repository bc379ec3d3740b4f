"""GPU parity at the FULL geometry (VideoMAE-v2-giant: 1408 wide, 40 blocks, all five heads) against the
golden vectors produced by the real reference (tests/golden/full_T16_all.npz).

L4P_F32 engine: 1e-3 relative-to-max on the sampled values (north_star).  L4P_BF16 engine: 40 residual blocks in bf16
cannot meet 1e-3 (SURVEY.md §7 "hard parts"); its drift is bounded at ~2x what is measured (rel-L2 of the samples:
encoder features 5e-3 -> gate 1e-2; heads / tracks / poses 0.6..1.2e-2 -> gate 3e-2).

test_batch4_*: the BENCHMARKED configuration (configs[2]: batch 4, bf16).  Batch 4 changes kernel selection (8-phase
256x256 GEMM / conv instead of 128x128, un-split attention, one tracker stream per clip), so it is tied to the batch-1
path clip by clip, and its first clip (the golden clip) to the reference's goldens.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd.models.utils import build_model
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch, sample_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]


@pytest.fixture(scope="module")
def full_sd():
    return seeded_state_dict(ModelCfg.full())


def _run(sd, precision):
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=precision)
    m.l4p_model.task_heads["camray"].use_intrinsics = True
    m.load_state_dict({"l4p_model." + k: v for k, v in sd.items()})
    batch = make_batch(16, 8)
    with torch.no_grad():
        out = m.forward({k: v.clone() for k, v in batch.items()}, ALL)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("precision", ["32-true", "bf16"])
def test_full_size_all_heads_vs_reference_goldens(dev, full_sd, precision):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T16_all.npz"))
    out = _run(full_sd, precision)
    feats = out["enc_features_bpc_2dlist"][0]
    report = {}
    for li in (14, 21, 28, 36, 40):
        f = feats.f32(li).float().cpu().reshape(-1)
        s, g = f[sample_indices(f.numel())], torch.from_numpy(gold[f"feat{li}"])
        report[f"feat{li}"] = (float((s - g).abs().max() / g.abs().max()), float((s - g).norm() / g.norm()))
    for k in gold.files:
        if k.startswith("feat") or k.startswith("trace"):
            continue
        y = out[k].float().cpu().reshape(-1)
        g = torch.from_numpy(gold[k]).reshape(-1)
        s = y[sample_indices(y.numel())] if y.numel() > 4096 else y
        report[k] = (float((s - g).abs().max() / g.abs().max()), float((s - g).norm() / g.norm()))
    print(precision, {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in report.items()})
    if precision == "32-true":
        bad = {k: v for k, v in report.items() if v[0] > 1e-3}
        assert not bad, bad
    else:
        bad = {k: v for k, v in report.items() if v[1] > bf16_gate(k)}
        assert not bad, bad


BF16_GATE_FEATURES, BF16_GATE_HEADS = 1e-2, 3e-2


def bf16_gate(key):
    return BF16_GATE_FEATURES if key.startswith("feat") else BF16_GATE_HEADS


OUT_KEYS = ["depth_est_b1thw", "flow_2d_backward_est_b2thw", "dyn_mask_est_b1thw", "traj3d_est_b16t", "track_2d_traj_est_bn2t",
            "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t"]


def _batch4():
    """configs[2]-shaped input: 4 different clips (clip 0 = the golden clip), 8 queries each."""
    bs = [make_batch(16, 8, seed=1234 + i) for i in range(4)]
    return {k: torch.cat([b[k] for b in bs], dim=0) for k in bs[0]}, bs


def _rel_l2(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def test_batch4_bf16_equals_four_batch1_forwards_and_goldens(dev, full_sd):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T16_all.npz"))
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision="bf16")
    m.l4p_model.task_heads["camray"].use_intrinsics = True
    m.load_state_dict({"l4p_model." + k: v for k, v in full_sd.items()})
    head = m.l4p_model.task_heads["track_2d"]
    b4, singles = _batch4()
    with torch.no_grad():
        head.trace = []
        out4 = m.forward({k: v.clone() for k, v in b4.items()}, ALL)
        trace4 = head.trace
        f4 = {li: out4["enc_features_bpc_2dlist"][0].f32(li).float().cpu() for li in (36, 40)}
        out4 = {k: out4[k].float().cpu() for k in OUT_KEYS}
        report, worst = {}, 0.0
        for i, b in enumerate(singles):
            head.trace = []
            o1 = m.forward({k: v.clone() for k, v in b.items()}, ALL)
            torch.cuda.synchronize()
            for li in (36, 40):
                f1 = o1["enc_features_bpc_2dlist"][0].f32(li).float().cpu()
                S = f1.shape[-2] if f1.dim() == 3 else f1.shape[0]
                report[(i, f"feat{li}")] = _rel_l2(f4[li].reshape(4, -1)[i], f1.reshape(-1))
            for k in OUT_KEYS:
                report[(i, k)] = _rel_l2(out4[k][i], o1[k].float().cpu()[0])
            # integer / boolean tracker state of clip i: identical between the two batch sizes
            t4 = trace4[i] if len(trace4) == 4 else None
            if t4 is not None:
                for name in ("labels", "prompt_labels", "valid_t"):
                    assert torch.equal(t4[name].cpu(), head.trace[0][name].cpu()), (i, name)
    print("B=4 vs B=1 rel-L2:", {f"{i}:{k}": f"{v:.2e}" for (i, k), v in report.items()})
    # same arithmetic, different tile shapes / summation order: differences at bf16 rounding level, well inside the
    # bf16-vs-f32 drift gates
    bad = {k: v for k, v in report.items() if v > (BF16_GATE_FEATURES if k[1].startswith("feat") else BF16_GATE_HEADS) / 2}
    assert not bad, bad
    # clip 0 of the batch against the reference's goldens, at the bf16 gates
    rep0 = {}
    for li in (36, 40):
        f = f4[li].reshape(4, -1)[0]
        g = torch.from_numpy(gold[f"feat{li}"])
        rep0[f"feat{li}"] = _rel_l2(f[sample_indices(f.numel())], g)
    for k in OUT_KEYS:
        y = out4[k][0].reshape(-1)
        g = torch.from_numpy(gold[k]).reshape(-1)
        rep0[k] = _rel_l2(y[sample_indices(y.numel())] if y.numel() > 4096 else y, g)
    print("B=4 clip 0 vs goldens:", {k: f"{v:.2e}" for k, v in rep0.items()})
    bad = {k: v for k, v in rep0.items() if v > bf16_gate(k)}
    assert not bad, bad


@pytest.mark.parametrize("precision", ["32-true", "bf16"])
def test_full_size_two_windows_vs_reference_goldens(dev, full_sd, precision):
    """The windowed path (configs[4]) at the REAL geometry: 24 frames = 2 overlapping windows through the reference itself
    (tools/gen_golden_full_windows.py -> tests/golden/full_T24_windows.npz): depth with the inverse-depth LstSq seam, backward
    flow, motion mask, and 4 tracks carried across the seam (memory tokens, re-seeding).  f32 engine 1e-3 relative-to-max;
    bf16 engine rel-L2 at the full-size gates of this file."""
    tasks = ["depth", "flow_2d_backward", "dyn_mask", "track_2d"]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T24_windows.npz"))
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=precision)
    m.load_state_dict({"l4p_model." + k: v for k, v in full_sd.items()})
    batch = make_batch(24, 4)
    with torch.no_grad():
        out = m.forward({k: v.clone() for k, v in batch.items()}, tasks)
    torch.cuda.synchronize()
    report = {}
    for k in gold.files:
        y = out[k].float().cpu().reshape(-1)
        g = torch.from_numpy(gold[k]).reshape(-1)
        s = y[sample_indices(y.numel())] if y.numel() > 4096 else y
        assert s.shape == g.shape, (k, tuple(s.shape), tuple(g.shape))
        report[k] = (float((s - g).abs().max() / g.abs().max()), float((s - g).norm() / g.norm()))
    print(report)
    for k, (emax, el2) in report.items():
        if precision == "32-true":
            assert emax <= 1e-3, (k, emax)
        else:
            assert el2 <= 3e-2, (k, el2)
