"""Seeded synthetic inputs + sampling helpers shared by tools/gen_golden.py and the parity tests.
(randn seed 1234 clip, fx=fy=224 / cx=cy=112 intrinsics, grid point queries.)"""
import torch

from l4p_amd.data.synthetic import grid_queries, synthetic_video  # noqa: F401  (shared with bench.py / demo.py)

QUERY_TIMES = [0, 0, 3, 10, 0, 18, 7, 25, 1, 12, 0, 21]


def make_batch(T: int, nq: int, seed: int = 1234):
    g = torch.Generator().manual_seed(seed)
    rgb = torch.randn([1, 3, T, 224, 224], generator=g, dtype=torch.float32)
    K = torch.eye(4, dtype=torch.float32)
    K[0, 0] = K[1, 1] = 224.0
    K[0, 2] = K[1, 2] = 112.0
    K = K[None, :, :, None].repeat(1, 1, 1, T)
    q = torch.zeros(1, nq, 3)
    for i in range(nq):
        t = min(QUERY_TIMES[i % len(QUERY_TIMES)], T - 2)
        q[0, i] = torch.tensor([t + 0.5, 14.0 + 28.0 * (i % 8) + 0.5, 14.0 + 28.0 * ((3 * i + 1) % 8) + 0.5])
    return {
        "rgb_b3thw": rgb,
        "intrinsics_b44t": K,
        "track_2d_pointquerries_bn3": q,
        "track_2d_pointlabels_bn": torch.ones(1, nq),
    }


LONG_QUERY_TIMES = [0, 0, 3, 10, 19, 27, 36, 44, 53, 61, 70, 82, 95, 107, 118, 130]


def long_batch(T: int = 136, seed: int = 4321):
    """The long-recursion case (tools/gen_golden_long.py -> tests/golden/mini_T136_long.npz): 16 windows, 16 tracks whose queries
    start anywhere in the video, so later windows see dead, newly started and re-seeded tracks side by side."""
    b = make_batch(T, len(LONG_QUERY_TIMES), seed)
    for i, t in enumerate(LONG_QUERY_TIMES):
        b["track_2d_pointquerries_bn3"][0, i, 0] = min(t, T - 2) + 0.5
    return b


def sample_indices(numel: int, n: int = 4096) -> torch.Tensor:
    if numel <= n:
        return torch.arange(numel)
    return torch.linspace(0, numel - 1, n, dtype=torch.float64).round().long()


# ---- clip preparation: seeded synthetic "decoded video" frames and the fixture cases ----------
PREPROCESS_CASES = {
    # up-scale 120x160 -> 224x224 and back (the blur is nearly the identity), mirror-pad 20 -> 39 frames, crop to 32
    "small_up": dict(seed=11, T=20, H=120, W=160, crop_size=(32, 224, 224), resize_size=(224, 224), max_frames=192,
                     stride=1, spacing=0.04),
    # down-scale 270x480 (support 1.2 / 2.1 taps), stride 2, crop_size None -> ceil(max(T,16)/8)*8 frames
    "down_stride": dict(seed=12, T=21, H=270, W=480, crop_size=None, resize_size=(224, 224), max_frames=192, stride=2,
                        spacing=0.1),
    # max_frames cuts the video (the reference keeps max_frames - 1 frames), odd sizes, non-square resize + centre crop
    "cut_nonsquare": dict(seed=13, T=12, H=135, W=241, crop_size=(16, 224, 224), resize_size=(298, 224), max_frames=10,
                          stride=1, spacing=0.25),
    # a single frame is repeated (l4p_dataset_mini.py:553-554)
    "single_frame": dict(seed=14, T=1, H=64, W=96, crop_size=(16, 224, 224), resize_size=(224, 224), max_frames=192,
                         stride=1, spacing=0.5),
}


def single_window_batch():
    """One 16-frame clip for the NON-windowed entry (always_use_windowed_version=False): queries at mixed times and point
    labels 0 / 1 / 2 given by the caller (the single-window forward must use them as they are)."""
    b = make_batch(16, 6)
    b["track_2d_pointlabels_bn"] = torch.tensor([[1.0, 0.0, 2.0, 1.0, 2.0, 0.0]])
    return b


def synthetic_rays(B=2, T=4, h=16, w=16, seed=5, noise=2e-3):
    """Pluecker ray maps [B,6,T,h,w] of cameras with ONE ray-grid K per batch item, rotations / centres per frame."""
    g = torch.Generator().manual_seed(seed)
    Ks, rays = [], torch.zeros(B, 6, T, h, w)
    j, i = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    pix = torch.stack([i, j, torch.ones_like(i)], dim=-1).reshape(-1, 3)
    for b in range(B):
        K = torch.tensor([[13.0 + b, 0.2, 7.3], [0.0, 14.5 - b, 7.9], [0.0, 0.0, 1.0]])
        Ks.append(K)
        for t in range(T):
            R = torch.linalg.qr(torch.eye(3) + 0.15 * torch.randn(3, 3, generator=g)).Q
            if torch.linalg.det(R) < 0:
                R = -R
            c = torch.randn(3, generator=g)
            d_cam = (torch.inverse(K) @ pix.T).T
            d_cam = d_cam / d_cam.norm(dim=-1, keepdim=True)
            d = d_cam @ R  # world-frame directions of cam_T_world rotation R: d_world = R^T d_cam
            d = d + noise * torch.randn(d.shape, generator=g)
            m = torch.cross(c.expand_as(d), d, dim=-1)
            rays[b, :3, t] = d.T.reshape(3, h, w)
            rays[b, 3:, t] = m.T.reshape(3, h, w)
    return rays, Ks


# ---- reference-anchored gates of the bf16 engine (round 4) ---------------------------------------------------------------------
BF16_MARGIN = 1.25        # per-key allowance over the reference's own figure (one run of each side; sampled values on ours)
BF16_MARGIN_SMALL = 2.0   # ... for outputs of fewer than 4096 values (4 tracks x 24 frames = 96 numbers: a rel-L2 over so few
                          # values from ONE run of each side is a noisy statistic; the geometric-mean bar below still holds)


def is_half(precision) -> bool:
    """The engine's IEEE-half mode (what the reference's shipped "16-mixed" computes in)."""
    return precision in ("16-mixed", "16-true", "f16", "fp16", 2)


def reference_autocast_drift(case: str, precision="bf16") -> dict:
    """tests/golden/reference_autocast_drift.json / ..._f16.json (tools/gen_golden_full_autocast.py [--dtype float16]): the imported
    reference under torch.autocast(bfloat16 / float16) against its own fp32 run on the golden inputs -> {key: rel-L2}."""
    import json
    import os

    name = "reference_autocast_drift_f16.json" if is_half(precision) else "reference_autocast_drift.json"
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)) as f:
        rep = json.load(f)[case]
    assert "autocast_failed" not in rep, rep
    return {k: v["rel_l2"] for k, v in rep.items() if isinstance(v, dict)} | {k: v for k, v in rep.items() if not isinstance(v, dict)}


def assert_bf16_within_reference_drift(report_l2: dict, case: str, keymap=None, what: str = "", small=(), precision="bf16") -> dict:
    """``report_l2``: {key: rel-L2 of the bf16 ENGINE against the reference's fp32 golden}.  The bar is the reference's own
    mixed-precision drift on the same inputs: every key within BF16_MARGIN x the reference's figure, and the geometric mean of
    engine / reference over the keys <= 1 (overall the engine is no further from the fp32 reference than the reference's own
    autocast run is).  Keys whose reference drift is 0 (pass-through values) must be exact to 1e-6.  ``small``: keys of outputs
    with fewer than 4096 values, held to BF16_MARGIN_SMALL per key.  Returns the ratios."""
    import math

    ref = reference_autocast_drift(case, precision)
    ratios = {}
    for k, v in report_l2.items():
        rk = (keymap or {}).get(k, k)
        assert rk in ref, (case, rk, sorted(ref))
        if ref[rk] == 0.0:
            assert v <= 1e-6, (what, k, v)
            continue
        ratios[k] = v / ref[rk]
    print(f"{'f16' if is_half(precision) else 'bf16'} engine drift / reference autocast drift [{case}{' ' + what if what else ''}]:",
          {k: f"{r:.2f}" for k, r in ratios.items()})
    bad = {k: r for k, r in ratios.items() if r > (BF16_MARGIN_SMALL if k in small else BF16_MARGIN)}
    assert not bad, (what, bad)
    gm = math.exp(sum(math.log(max(r, 1e-12)) for r in ratios.values()) / max(len(ratios), 1))
    assert gm <= 1.0, (what, gm, ratios)
    return ratios


def integer_state_mismatches(trace, gold, nwin: int, sel=slice(None)) -> torch.Tensor:
    """Engine trace (VideoMAETrack2DSamHead.trace of ONE clip, window order) against the ``trace{w}_*`` arrays of a golden file
    (labels / prompt labels / query times from the reference's own run; valid_t / best_vis_id from the oracle on the reference's
    features, asserted equal to the reference where both exist).  -> bool per track: state differs anywhere in the recursion."""
    import numpy as np

    assert len(trace) == nwin, (len(trace), nwin)
    n = trace[0]["labels"].numel()
    bad = torch.zeros(n, dtype=torch.bool)
    for w in range(nwin):
        tr = trace[w]
        assert tr["window"] == w
        bad |= torch.from_numpy(tr["labels"].cpu().numpy() != gold[f"trace{w}_labels"][sel])
        bad |= torch.from_numpy(tr["prompt_labels"].cpu().numpy() != gold[f"trace{w}_prompt_labels"][sel])
        bad |= torch.from_numpy(tr["queries"][:, 0].cpu().numpy() != gold[f"trace{w}_queries"][sel][:, 0])
        if f"trace{w}_valid_t" in gold:
            bad |= torch.from_numpy((tr["valid_t"].cpu().numpy().astype(bool) != gold[f"trace{w}_valid_t"][sel]).any(axis=-1))
        if f"trace{w}_best_vis_id" in gold and "best_vis_id" in tr:
            bad |= torch.from_numpy(tr["best_vis_id"].cpu().numpy().astype(np.int64) != gold[f"trace{w}_best_vis_id"][sel])
    return bad
