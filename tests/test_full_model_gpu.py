"""GPU parity at the FULL geometry (VideoMAE-v2-giant: 1408 wide, 40 blocks, all five heads) against the
golden vectors produced by the real reference (tests/golden/full_T16_all.npz).

L4P_F32 engine: 1e-3 relative-to-max on the sampled values (north_star).  L4P_BF16 engine (what bench.py times): 40 residual
blocks in bf16 cannot meet 1e-3 (SURVEY.md §7 "hard parts") - and neither can the reference's own mixed-precision mode (its demo
runs "16-mixed", demo.py:22-23).  The bar is therefore the REFERENCE'S OWN DRIFT: the imported reference under
torch.autocast(bfloat16) against its own fp32 run on the same inputs (tools/gen_golden_full_autocast.py ->
tests/golden/reference_autocast_drift.json: features 0.9..1.3e-2, depth 1.7e-2, flow / mask 0.9..1.0e-2, poses 1.1e-2, tracks
0.2..1.0e-2 rel-L2).  The engine's rel-L2 against the fp32 goldens must stay within it key by key (golden_utils.
assert_bf16_within_reference_drift: <= 1.25 x per key, geometric mean of the ratios <= 1); measured ratios 0.4..1.0.
Integer / boolean tracker state (labels, prompt labels, validity masks, re-seeded query times, argmax index) is asserted
BIT-EXACT against the reference's recorded trace for the f32 engine at this geometry over 1, 2 and 4 windows; for the bf16 engine
the tracks whose state differs are counted and bounded by the count the reference's own autocast run shows (min. 1).

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
from tests.golden_utils import (assert_bf16_within_reference_drift, grid_queries, integer_state_mismatches, make_batch,
                                reference_autocast_drift, sample_indices)

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
    head = m.l4p_model.task_heads["track_2d"]
    head.trace = []
    with torch.no_grad():
        out = m.forward({k: v.clone() for k, v in batch.items()}, ALL)
    torch.cuda.synchronize()
    return out, head.trace


def _check_integer_state(trace, gold, nwin, precision, case, sel=slice(None), suffix=""):
    """f32 engine: bit-exact against the reference's trace.  bf16 engine: tracks whose state differs anywhere in the recursion,
    bounded by what the reference's own autocast run shows on these inputs (at least one near-tie is allowed)."""
    bad = integer_state_mismatches(trace, gold, nwin, sel)
    if precision == "32-true":
        assert not bool(bad.any()), (case, bad)
        return
    ref_count = int(reference_autocast_drift(case, precision)["tracks_with_differing_integer_state" + suffix])
    print(f"bf16 integer-state mismatches vs the reference's trace ({case}): {int(bad.sum())} of {bad.numel()} tracks "
          f"(the reference's own autocast run: {ref_count})")
    assert int(bad.sum()) <= max(1, ref_count), (case, bad)


@pytest.mark.parametrize("precision", ["32-true", "bf16", "16-mixed"])
def test_full_size_all_heads_vs_reference_goldens(dev, full_sd, precision):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T16_all.npz"))
    out, trace = _run(full_sd, precision)
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
        assert_bf16_within_reference_drift({k: v[1] for k, v in report.items()}, "full_T16_all", precision=precision)
    _check_integer_state(trace, gold, 1, precision, "full_T16_all")


def _self_gate(report, case="full_T16_all", track_case=None):
    """Two bf16 runs of the engine through different kernels (batch sizes): their distance is gated by the reference's own
    autocast drift of that output (each run is within that of the fp32 result).  ``track_case``: the fixture whose query set the
    tracks belong to."""
    ref, tref = reference_autocast_drift(case), reference_autocast_drift(track_case or case)
    return {k: v for k, v in report.items() if v > (tref if k[1].startswith("track_2d") else ref)[k[1]]}


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
            assert len(trace4) == 4 and trace4[i]["clip"] == i and len(head.trace) == 1, (len(trace4), len(head.trace))
            for name in ("labels", "prompt_labels", "valid_t"):
                assert torch.equal(trace4[i][name].cpu(), head.trace[0][name].cpu()), (i, name)
    print("B=4 vs B=1 rel-L2:", {f"{i}:{k}": f"{v:.2e}" for (i, k), v in report.items()})
    # same arithmetic, different tile shapes / summation order: differences at bf16 rounding level, well inside the
    # bf16-vs-f32 drift gates
    bad = _self_gate(report)
    assert not bad, bad
    # clip 0 of the batch against the reference's goldens, within the reference's own autocast drift
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
    assert_bf16_within_reference_drift(rep0, "full_T16_all", what="batch 4, clip 0")
    _check_integer_state([trace4[0]], gold, 1, "bf16", "full_T16_all")


def test_batch8_per_gpu_batch_of_configs3(dev, full_sd):
    """configs[3] runs 8 clips per GPU (what `bench.py --gpus N` gives every rank): batch 8 selects other tile forms than batch 4
    (256 x 192 tiles for the N = 1408 projections at 512 tiles, 4 attention tiles per persistent workgroup, 1.5-round GEMMs).
    All heads, bf16: clip 0 (the golden clip) against the reference's goldens at the bf16 gates, clips 3 and 7 against their own
    batch-1 forwards (integer tracker state identical, floats at rounding level)."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T16_all.npz"))
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision="bf16")
    m.l4p_model.task_heads["camray"].use_intrinsics = True
    m.load_state_dict({"l4p_model." + k: v for k, v in full_sd.items()})
    head = m.l4p_model.task_heads["track_2d"]
    bs = [make_batch(16, 8, seed=1234 + i) for i in range(8)]
    b8 = {k: torch.cat([b[k] for b in bs], dim=0) for k in bs[0]}
    with torch.no_grad():
        head.trace = []
        out8 = m.forward({k: v.clone() for k, v in b8.items()}, ALL)
        trace8 = head.trace
        f8 = {li: out8["enc_features_bpc_2dlist"][0].f32(li).float().cpu() for li in (36, 40)}
        out8 = {k: out8[k].float().cpu() for k in OUT_KEYS}
        assert len(trace8) == 8
        report = {}
        for i in (3, 7):
            head.trace = []
            o1 = m.forward({k: v.clone() for k, v in bs[i].items()}, ALL)
            torch.cuda.synchronize()
            for li in (36, 40):
                f1 = o1["enc_features_bpc_2dlist"][0].f32(li).float().cpu()
                report[(i, f"feat{li}")] = _rel_l2(f8[li].reshape(8, -1)[i], f1.reshape(-1))
            for k in OUT_KEYS:
                report[(i, k)] = _rel_l2(out8[k][i], o1[k].float().cpu()[0])
            assert trace8[i]["clip"] == i and len(head.trace) == 1
            for name in ("labels", "prompt_labels", "valid_t"):
                assert torch.equal(trace8[i][name].cpu(), head.trace[0][name].cpu()), (i, name)
    head.trace = None
    print("B=8 vs B=1 rel-L2:", {f"{i}:{k}": f"{v:.2e}" for (i, k), v in report.items()})
    bad = _self_gate(report)
    assert not bad, bad
    rep0 = {}
    for li in (36, 40):
        f = f8[li].reshape(8, -1)[0]
        rep0[f"feat{li}"] = _rel_l2(f[sample_indices(f.numel())], torch.from_numpy(gold[f"feat{li}"]))
    for k in OUT_KEYS:
        y = out8[k][0].reshape(-1)
        g = torch.from_numpy(gold[k]).reshape(-1)
        rep0[k] = _rel_l2(y[sample_indices(y.numel())] if y.numel() > 4096 else y, g)
    print("B=8 clip 0 vs goldens:", {k: f"{v:.2e}" for k, v in rep0.items()})
    assert_bf16_within_reference_drift(rep0, "full_T16_all", what="batch 8, clip 0")


@pytest.mark.parametrize("precision", ["32-true", "bf16", "16-mixed"])
def test_full_size_two_windows_vs_reference_goldens(dev, full_sd, precision):
    """The windowed path (configs[4]) at the REAL geometry: 24 frames = 2 overlapping windows through the reference itself
    (tools/gen_golden_full_windows.py -> tests/golden/full_T24_windows.npz): depth with the inverse-depth LstSq seam, backward
    flow, motion mask, and 4 tracks carried across the seam (memory tokens, re-seeding).  f32 engine 1e-3 relative-to-max;
    bf16 engine within the reference's own autocast drift on these inputs; the tracker's integer state across the seam against
    the reference's recorded trace (bit-exact in f32)."""
    tasks = ["depth", "flow_2d_backward", "dyn_mask", "track_2d"]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T24_windows.npz"))
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=precision)
    m.load_state_dict({"l4p_model." + k: v for k, v in full_sd.items()})
    batch = make_batch(24, 4)
    head = m.l4p_model.task_heads["track_2d"]
    head.trace = []
    with torch.no_grad():
        out = m.forward({k: v.clone() for k, v in batch.items()}, tasks)
    torch.cuda.synchronize()
    report = {}
    for k in gold.files:
        if k.startswith("trace"):
            continue
        y = out[k].float().cpu().reshape(-1)
        g = torch.from_numpy(gold[k]).reshape(-1)
        s = y[sample_indices(y.numel())] if y.numel() > 4096 else y
        assert s.shape == g.shape, (k, tuple(s.shape), tuple(g.shape))
        report[k] = (float((s - g).abs().max() / g.abs().max()), float((s - g).norm() / g.norm()))
    print(report)
    if precision == "32-true":
        for k, (emax, el2) in report.items():
            assert emax <= 1e-3, (k, emax)
    else:
        assert_bf16_within_reference_drift({k: v[1] for k, v in report.items()}, "full_T24_windows",
                                           small=[k for k in report if out[k].numel() < 4096], precision=precision)
    # integer / boolean tracker state over the seam, at the real geometry, against the reference's own trace
    _check_integer_state(head.trace, gold, 2, precision, "full_T24_windows")


def _samples(y, g):
    y = y.float().cpu().reshape(-1)
    g = torch.from_numpy(np.asarray(g)).reshape(-1)
    s = y[sample_indices(y.numel())] if y.numel() > 4096 else y
    assert s.shape == g.shape, (tuple(s.shape), tuple(g.shape))
    return s, g


def test_benchmarked_configuration_itself(dev, full_sd):
    """configs[2] EXACTLY as bench.py runs it: batch 4, 64 grid queries per clip (golden_utils.grid_queries), the SHIPPED camray
    configuration (configs/model.yaml:44-45 use_intrinsics=false: K estimated from the first window's ray map), one tracker
    stream per clip with the join deferred behind the dense decoders, the integer state traced WITHOUT touching the stream
    schedule.  Checked:
      * every clip against its own batch-1 forward (different kernels), integer / boolean tracker state identical;
      * clip 0 (the golden clip) against the REFERENCE: dense outputs vs full_T16_all.npz, its 64 tracks vs full_T16_q64.npz
        (tools/gen_golden_full_joint.py: the reference's tracker on the benchmark's query set);
      * the K estimate of every clip against the oracle's restatement of the engine's estimator on the same ray map, and the
        poses against the reference flow downstream of that estimate (oracle rays_to_cameras_fixed_intrinsics, k_override)."""
    from oracle import l4p_oracle as lo

    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T16_all.npz"))
    gq = np.load(os.path.join(ROOT, "tests", "golden", "full_T16_q64.npz"))
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision="bf16")
    net = m.l4p_model
    assert net.task_heads["camray"].use_intrinsics is False and net.task_heads["camray"].fixed_intrinsics is True  # as shipped
    m.load_state_dict({"l4p_model." + k: v for k, v in full_sd.items()})
    head = net.task_heads["track_2d"]
    singles = []
    for i in range(4):
        b = make_batch(16, 1, seed=1234 + i)
        b["track_2d_pointquerries_bn3"] = grid_queries(64)
        b["track_2d_pointlabels_bn"] = torch.ones(1, 64)
        singles.append(b)
    b4 = {k: torch.cat([b[k] for b in singles], dim=0) for k in singles[0]}
    keys = OUT_KEYS + ["traj3d_intrinsics_est_b16t"]
    os.environ.pop("L4P_TRACK_STREAMS", None)
    with torch.no_grad():
        head.trace = []
        out4 = m.forward({k: v.clone() for k, v in b4.items()}, ALL)
        torch.cuda.synchronize()
        trace4 = head.trace
        assert [(t["clip"], t["window"]) for t in trace4] == [(i, 0) for i in range(4)]
        assert len(getattr(head, "_clip_streams", [])) >= 4, "the clips' trackers did not run on their own streams"
        out4 = {k: out4[k].float().cpu() for k in keys}
        # the ray maps the forward decoded (deterministic: the same launches again)
        data = {k: v.to(net.device) for k, v in b4.items()}
        rays4 = net.task_heads["camray"]._decode(net.encode_features(data, ALL), (16, 224, 224)).float().cpu()
        report = {}
        for i, b in enumerate(singles):
            head.trace = []
            o1 = m.forward({k: v.clone() for k, v in b.items()}, ALL)
            torch.cuda.synchronize()
            for k in keys:
                report[(i, k)] = _rel_l2(out4[k][i], o1[k].float().cpu()[0])
            for name in ("labels", "prompt_labels", "valid_t"):
                assert torch.equal(trace4[i][name].cpu(), head.trace[0][name].cpu()), (i, name)
            assert torch.equal(trace4[i]["queries"][:, 0].cpu(), head.trace[0]["queries"][:, 0].cpu()), i
        head.trace = None
    print("bench config, B=4 vs B=1 rel-L2:", {f"{i}:{k}": f"{v:.2e}" for (i, k), v in report.items()})
    # The K estimate is a consensus (best-of-128 + refit) estimate on a ray map that, with random weights, is not the image
    # of any camera: the batch-4 and batch-1 ray maps differ by bf16 rounding and the estimate legitimately lands elsewhere
    # (measured: rel-L2 ~1 between the two K).  Poses / K are therefore NOT tied to the batch-1 path; they are tied, for every
    # clip, to the CPU restatement of the estimator on the very ray map the batch-4 forward decoded (below).
    bad = _self_gate({k: v for k, v in report.items() if not k[1].startswith("traj3d")}, track_case="full_T16_q64")
    assert not bad, bad
    # ---- clip 0 against the reference ----
    rep0 = {}
    for k in ("depth_est_b1thw", "flow_2d_backward_est_b2thw", "dyn_mask_est_b1thw"):
        rep0[k] = _rel_l2(*_samples(out4[k][0], gold[k]))
    for k in ("track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t"):
        assert tuple(out4[k][0].shape) == tuple(gq[k][0].shape), k
        rep0[k] = _rel_l2(out4[k][0], torch.from_numpy(gq[k][0]))
    print("bench config, clip 0 vs the reference:", {k: f"{v:.2e}" for k, v in rep0.items()})
    assert_bf16_within_reference_drift({k: v for k, v in rep0.items() if not k.startswith("track")}, "full_T16_all",
                                       what="bench config, dense")
    assert_bf16_within_reference_drift({k: v for k, v in rep0.items() if k.startswith("track")}, "full_T16_q64",
                                       what="bench config, 64 grid queries")
    # ---- K estimate and what follows from it, every clip ----
    for i in range(4):
        dirs = rays4[i, :3, 0].reshape(3, -1).T.numpy()
        want, n_cons, iters = lo.engine_rays_to_intrinsics(dirs, 16, 16, 224, 224, thr=0.2, b=i)
        K_pix = out4["traj3d_intrinsics_est_b16t"][i].reshape(4, 4, 16)
        assert torch.equal(K_pix[..., 0], K_pix[..., 15])
        got = K_pix[..., 0].double().numpy()
        print(f"clip {i}: consensus {n_cons}/256 after {iters} rounds, fx {got[0, 0]:.2f} fy {got[1, 1]:.2f}")
        # (kernel == restatement to 1e-4 on camera-like ray maps with noise and gross outliers, tests/test_intrinsics_gpu.py.  With
        #  random weights the ray map is the image of NO camera: the consensus is 8..13 of 256 rays — the estimator's floor —,
        #  the 9x9 DLT system on them is ill-conditioned and the two evaluations of the same schedule drift apart (measured: 5e-3
        #  at 12 rays, 0.36 at 8).  The comparison is therefore gated only where a consensus exists; what the reference does WITH
        #  the estimate is checked for every clip below.)
        if n_cons >= 32:
            assert np.abs(got - want).max() <= 2e-2 * np.abs(want).max(), (i, got, want)
        K_ray = lo.denormalize_intrinsics(lo.normalize_intrinsics(K_pix[None], 224, 224), 16, 16)[0, :3, :3, 0]
        E, Kout = lo.rays_to_cameras_fixed_intrinsics(rays4[i:i + 1], (224, 224), k_override=lambda b: K_ray)
        pose = torch.linalg.inv(E.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).reshape(16, 16)
        y = out4["traj3d_est_b16t"][i]
        assert (y - pose).abs().max() <= 1e-3 * pose.abs().max(), (i, float((y - pose).abs().max() / pose.abs().max()))
        assert (Kout[0].reshape(16, 16) - out4["traj3d_intrinsics_est_b16t"][i]).abs().max() <= 1e-3 * Kout.abs().max()
    # ---- the estimator's CONSENSUS branch in this very configuration (batch 4, clip streams on, shipped use_intrinsics=false): with
    #      random weights no clip's ray map has a consensus, so the comparison above never runs.  Clip 3's decoded ray map is
    #      replaced AT THE HEAD'S OUTPUT (after the real decoder kernels have run, on their stream) by the rendering of a known
    #      camera path with measurement noise (tests/test_wellposed_heads_gpu._scene, window 0), and the same forward runs again.
    from tests.test_wellposed_heads_gpu import _scene

    K_true, _, _, wins = _scene()
    rays_wp = wins[0]["rays"]
    cam = net.task_heads["camray"]
    orig_decode = cam._decode
    assert tuple(rays_wp.shape) == (1, 6, 16, 16, 16)

    def decode_with_rendered_clip3(feats, img_info):
        r = orig_decode(feats, img_info)
        assert tuple(r.shape) == (4, 6, 16, 16, 16), r.shape
        r[3].copy_(rays_wp[0].to(r.device, r.dtype))
        return r

    cam._decode = decode_with_rendered_clip3
    try:
        with torch.no_grad():
            out_wp = m.forward({k: v.clone() for k, v in b4.items()}, ALL)
        torch.cuda.synchronize()
    finally:
        del cam._decode
    assert len(getattr(head, "_clip_streams", [])) >= 4
    dirs = rays_wp[0, :3, 0].reshape(3, -1).T.numpy()
    want, n_cons, iters = lo.engine_rays_to_intrinsics(dirs, 16, 16, 224, 224, thr=0.2, b=3)
    assert n_cons >= 200, n_cons  # a real consensus: the gated comparison executes
    K_pix = out_wp["traj3d_intrinsics_est_b16t"].float().cpu()[3].reshape(4, 4, 16)
    got = K_pix[..., 0].double().numpy()
    print(f"well-posed clip 3 at batch 4: consensus {n_cons}/256 after {iters} rounds, fx {got[0, 0]:.2f} fy {got[1, 1]:.2f} "
          f"(true {float(K_true[0, 0]):.1f} {float(K_true[1, 1]):.1f})")
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max(), (got, want)           # kernel == its CPU restatement
    assert np.abs(got[:3, :3] - K_true[:3, :3].double().numpy()).max() <= 2e-2 * float(K_true[0, 0]), got  # and the true camera
    # the other clips' K do not depend on clip 3's ray map (per-clip estimate, hashed per clip)
    for i in range(3):
        assert torch.equal(out_wp["traj3d_intrinsics_est_b16t"].float().cpu()[i], out4["traj3d_intrinsics_est_b16t"][i]), i
    # poses of clip 3: the reference flow downstream of the estimate, and the rendered camera path itself
    K_ray = lo.denormalize_intrinsics(lo.normalize_intrinsics(K_pix[None], 224, 224), 16, 16)[0, :3, :3, 0]
    E, _ = lo.rays_to_cameras_fixed_intrinsics(rays_wp, (224, 224), k_override=lambda b: K_ray)
    pose = torch.linalg.inv(E.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).reshape(16, 16)
    y = out_wp["traj3d_est_b16t"].float().cpu()[3]
    assert (y - pose).abs().max() <= 1e-3 * pose.abs().max(), float((y - pose).abs().max() / pose.abs().max())
    rel = wins[0]["rel"].reshape(16, 16).float()
    assert (y - rel).abs().max() <= 5e-2, float((y - rel).abs().max())  # (measurement noise of the rendering; K to 2 %)


@pytest.mark.parametrize("precision", ["32-true", "bf16", "16-mixed"])
def test_depth_only_full_size_vs_reference_golden(dev, precision):
    """configs[1] as bench.py --workload c2 builds it: a model with the depth head only, task list ["depth"], batch 1.  The
    encoder stops after block 36 (the highest hook; the reference runs all 40 and discards them) and the batch-1 dispatch
    (split-K MLP-out projection, KV-split attention) runs; the output must equal the reference's depth_est_b1thw."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T16_all.npz"))
    cfg = ModelCfg.full()
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=precision)
    net = m.l4p_model
    net.task_heads = torch.nn.ModuleDict({"depth": net.task_heads["depth"]})
    m.load_state_dict({"l4p_model." + k: v for k, v in seeded_state_dict(cfg, tasks=["depth"]).items()})
    batch = make_batch(16, 8)
    with torch.no_grad():
        out = m.forward({k: v.clone() for k, v in batch.items()}, ["depth"])
    torch.cuda.synchronize()
    feats = out["enc_features_bpc_2dlist"][0]
    with pytest.raises(KeyError):
        feats.f32(cfg.depth)  # blocks 37..40 never ran
    s, g = _samples(out["depth_est_b1thw"], gold["depth_est_b1thw"])
    emax, el2 = float((s - g).abs().max() / g.abs().max()), float((s - g).norm() / g.norm())
    print(precision, "depth-only vs reference:", f"max {emax:.2e} rel-L2 {el2:.2e}")
    if precision == "32-true":
        assert emax <= 1e-3, emax
    else:
        assert_bf16_within_reference_drift({"depth_est_b1thw": el2}, "full_T16_all", what="depth only", precision=precision)


@pytest.mark.parametrize("precision", ["32-true", "bf16", "16-mixed"])
def test_full_size_four_windows_joint_vs_reference_goldens(dev, full_sd, precision):
    """The path of dense_heads.py:360-492 at the REAL geometry over more than one seam: 40 frames = 4 windows = 3 seams, all
    five tasks, 8 tracks (tools/gen_golden_full_joint.py -> tests/golden/full_T40_joint.npz).
      * flow, motion mask, tracks: against the reference's outputs directly;
      * depth / poses / K of frames 0..7 (first window, never re-aligned): against the reference directly;
      * the jointly aligned depth / poses over all 40 frames: the reference's two random draws cannot be pinned, so the fixture
        carries, next to the reference run with fixed stand-ins (which pins the oracle's flow over 3 seams at this geometry),
        the oracle's flow with the ENGINE's deterministic draws ("engine.*") — the f32 engine must reproduce it to 1e-3;
        the bf16 engine's joint stage is checked on its own per-window estimates (as tests/test_joint_gpu.py does at the mini
        geometry: a best-of-100 estimator legitimately lands elsewhere for inputs that differ by the bf16 drift)."""
    from oracle import joint_oracle as jo

    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T40_joint.npz"))
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=precision)
    m.l4p_model.task_heads["camray"].use_intrinsics = True
    m.load_state_dict({"l4p_model." + k: v for k, v in full_sd.items()})
    batch = make_batch(40, 8)
    head = m.l4p_model.task_heads["track_2d"]
    head.trace = []
    with torch.no_grad():
        out = m.forward({k: v.clone() for k, v in batch.items()}, ALL)
    torch.cuda.synchronize()
    trace = head.trace
    head.trace = None
    exact = precision == "32-true"
    # integer / boolean tracker state over 4 windows / 3 re-seedings at the real geometry, against the reference's own trace
    _check_integer_state(trace, gold, 4, precision, "full_T40_track24", suffix="[:8]")
    report = {}
    for k in ("flow_2d_backward_est_b2thw", "dyn_mask_est_b1thw", "track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t",
              "track_2d_depth_est_bn1t"):
        s, g = _samples(out[k], gold[k])
        report[k] = (float((s - g).abs().max() / g.abs().max()), float((s - g).norm() / g.norm()))
    joint_keys = ("depth_est_b1thw", "traj3d_est_b16t", "traj3d_intrinsics_est_b16t")
    for k in joint_keys:
        s, g = _samples(out[k], gold["engine." + k])
        report["engine." + k] = (float((s - g).abs().max() / g.abs().max()), float((s - g).norm() / g.norm()))
    print(precision, {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in report.items()})
    if exact:
        for k, (emax, el2) in report.items():
            assert emax <= 1e-3, (k, emax)
    else:
        # the reference's own autocast drift: the stitched flow / mask from its 2-window run, the tracks from its 4-window run of
        # these very queries (the first 8 of full_T40_track24's)
        assert_bf16_within_reference_drift({k: report[k][1] for k in ("flow_2d_backward_est_b2thw", "dyn_mask_est_b1thw")},
                                           "full_T24_windows", what="4 windows, dense", precision=precision)
        assert_bf16_within_reference_drift({k: report[k][1] for k in report if k.startswith("track_2d")}, "full_T40_track24",
                                           keymap={k: k + "[:8]" for k in report}, what="4 windows, tracks", precision=precision,
                                           small=[k for k in report if k.startswith("track_2d") and out[k].numel() < 4096])  # (8 tracks x 40 frames)
    # frames 0..7 are written by window 0 only: independent of the draws -> the reference's own values (full tensors are not
    # in the fixture; the sampled positions that fall into the first 8 frames are compared)
    first_win = {}
    for k in joint_keys:
        y = out[k].float().cpu()
        idx = sample_indices(y.numel()) if y.numel() > 4096 else torch.arange(y.numel())
        t_of = (idx // (224 * 224)) % 40 if k == "depth_est_b1thw" else idx % 40
        sel = t_of < 8
        s, g = y.reshape(-1)[idx][sel], torch.from_numpy(gold[k]).reshape(-1)[sel]
        assert int(sel.sum()) > 10
        e = float((s - g).abs().max() / g.abs().max()) if exact else float((s - g).norm() / g.norm())
        if exact:
            assert e <= 1e-3, ("first window", k, e)
        else:
            first_win[k] = e
    if exact:
        return
    # (one gate over both outputs; the pose sample is 16 x 8 numbers: the small-sample margin)
    assert_bf16_within_reference_drift(first_win, "full_T16_all", what="4 windows, first window", precision=precision,
                                       small=[k for k in first_win if k.startswith("traj3d")])
    per_win = []
    with torch.no_grad():
        for st in (0, 8, 16, 24):
            b = {k: (v[:, :, st:st + 16].clone() if k == "rgb_b3thw" else v[..., st:st + 16].clone() if k == "intrinsics_b44t" else v.clone())
                 for k, v in batch.items()}
            o = m.forward(b, ["depth", "camray"])
            per_win.append({"depth": o["depth_est_b1thw"].float().cpu(), "camray": o["traj3d_est_b16t"].float().cpu(),
                            "camray_intrinsics_est": o["traj3d_intrinsics_est_b16t"].float().cpu()})
    torch.cuda.synchronize()
    log = []
    est = jo.joint_windowed(lambda w: {k: v.clone() for k, v in per_win[w].items()}, [0, 8, 16, 24], 16, "engine", log)
    for key, ek in (("depth_est_b1thw", "depth"), ("traj3d_est_b16t", "camray"), ("traj3d_intrinsics_est_b16t", "camray_intrinsics_est")):
        y, r = out[key].float().cpu(), est[ek]
        assert (y - r).abs().max() <= 1e-3 * r.abs().max(), (key, float((y - r).abs().max() / r.abs().max()), log)


@pytest.mark.parametrize("precision", ["32-true", "bf16", "16-mixed"])
def test_full_size_24_tracks_four_windows_integer_state(dev, full_sd, precision):
    """24 tracks with mixed start frames over 40 frames = 4 windows = 3 memory updates / re-seedings, tracker only, at the real
    geometry, against the reference's own run (tools/gen_golden_full_autocast.py -> tests/golden/full_T40_track24.npz: complete
    outputs + the per-window integer / boolean state).  f32 engine: floats 1e-3 relative-to-max, state bit-exact.  bf16 engine:
    the reference's own autocast run flips the state of 3 of these 24 tracks (a re-seed argmax over 8 visibilities is a
    near-tie for some): the engine is bounded by that count, and on the tracks whose state it reproduces its floats are within
    the reference's drift."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "full_T40_track24.npz"))
    m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=precision)
    m.load_state_dict({"l4p_model." + k: v for k, v in full_sd.items()})
    head = m.l4p_model.task_heads["track_2d"]
    head.trace = []
    batch = make_batch(40, 24)
    with torch.no_grad():
        out = m.forward({k: v.clone() for k, v in batch.items()}, ["track_2d"])
    torch.cuda.synchronize()
    trace, head.trace = head.trace, None
    _check_integer_state(trace, gold, 4, precision, "full_T40_track24")
    same = ~integer_state_mismatches(trace, gold, 4)
    rep = {}
    for k in ("track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t"):
        y, g = out[k].float().cpu(), torch.from_numpy(gold[k])
        assert y.shape == g.shape, k
        if precision == "32-true":
            assert (y - g).abs().max() <= 1e-3 * g.abs().max(), (k, float((y - g).abs().max() / g.abs().max()))
        else:
            rep[k] = _rel_l2(y[:, same], g[:, same])
    if rep:
        print("24 tracks, 4 windows, bf16, tracks with the reference's state:", int(same.sum()), {k: f"{v:.2e}" for k, v in rep.items()})
        assert_bf16_within_reference_drift(rep, "full_T40_track24", what="24 tracks", precision=precision)


def test_folded_value_projection_equals_projected_values_at_full_size(dev, full_sd, monkeypatch):
    """The tracker at the FULL geometry (1408 channels, 8 heads of 88: where the value projection of the token -> image attentions is
    folded away - l4p_t2i_probs / l4p_t2i_context; the mini geometry's head dim 44 keeps the projected values) against
    the projected form (L4P_TRACK_FOLD_T2I_V=0) over 3 windows (24 frames) with 8 tracks: the same function of the same weights.
    f32 engine: equal to rounding (1e-4 of the maximum), values that mark invalid entries identical.  bf16 engine: each form as close
    to the f32 engine as the other (mean distance of the tracks within 1.5x, worst track within 2x), and the native window call bit-identical to the Python composition
    that the switch is read by."""
    batch = make_batch(24, 8)
    keys = ["track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t"]
    res = {}
    for precision in ("32-true", "bf16"):
        m = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=precision)
        m.load_state_dict({"l4p_model." + k: v for k, v in full_sd.items()})
        with torch.no_grad():
            monkeypatch.setenv("L4P_TRACK_PYTHON", "1")
            monkeypatch.setenv("L4P_TRACK_FOLD_T2I_V", "1")
            a = m.forward({k: v.clone() for k, v in batch.items()}, ["track_2d"])
            monkeypatch.setenv("L4P_TRACK_FOLD_T2I_V", "0")
            b = m.forward({k: v.clone() for k, v in batch.items()}, ["track_2d"])
            monkeypatch.delenv("L4P_TRACK_PYTHON")
            monkeypatch.delenv("L4P_TRACK_FOLD_T2I_V")
            c = m.forward({k: v.clone() for k, v in batch.items()}, ["track_2d"])
        torch.cuda.synchronize()
        res[precision] = ({k: a[k].float().cpu() for k in keys}, {k: b[k].float().cpu() for k in keys})
        for k in keys:
            assert torch.equal(a[k], c[k]), (precision, k, "native vs Python composition", float((a[k] - c[k]).abs().max()))
        del m

    def per_track(x, y):
        return (x - y)[0].flatten(1).norm(dim=1) / y[0].flatten(1).norm(dim=1).clamp_min(1e-9)

    for k in keys:
        fa, fb = res["32-true"][0][k], res["32-true"][1][k]
        ha, hb = res["bf16"][0][k], res["bf16"][1][k]
        err = float((fa - fb).abs().max() / fb.abs().max())
        d_fold, d_proj = per_track(ha, fb), per_track(hb, fb)
        print(k, f"f32 folded vs projected values: max {err:.2e}; bf16 vs f32 per track (mean, max): folded {float(d_fold.mean()):.1e} "
                 f"{float(d_fold.max()):.1e}, projected {float(d_proj.mean()):.1e} {float(d_proj.max()):.1e}")
        assert err <= 1e-4, (k, err)
        assert torch.equal(fa == -10.0, fb == -10.0) and torch.equal(fa == 0.0, fb == 0.0), k
        # (8 tracks, two evaluation orders in bf16: single tracks move either way by a factor of two - see the printed rows -, so the
        #  forms are compared on the tracks' mean and worst distance, not track by track)
        assert float(d_fold.mean()) <= 1.5 * float(d_proj.mean()) + 1e-4, (k, d_fold, d_proj)
        assert float(d_fold.max()) <= 2.0 * float(d_proj.max()) + 1e-3, (k, d_fold, d_proj)
