"""CPU: the oracle (oracle/l4p_oracle.py) against the golden vectors produced by the REAL reference
(tools/gen_golden.py; tests/golden/*.npz).  Tolerance 1e-4 relative-to-max (measured <= 1e-6);
tracker labels / prompt labels / re-seeded query times bit-exact."""
import os

import numpy as np
import pytest
import torch

from l4p_amd.weights import ModelCfg, seeded_state_dict
from oracle.l4p_oracle import OracleModel, encoder_forward
from tests.golden_utils import make_batch, sample_indices

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]


@pytest.fixture(scope="module")
def mini():
    cfg = ModelCfg.mini()
    return cfg, seeded_state_dict(cfg)


def _cmp(name, y, g):
    y = y.detach().float().reshape(-1)
    g = torch.from_numpy(np.asarray(g)).float().reshape(-1)
    s = y[sample_indices(y.numel())] if y.numel() > 4096 else y
    assert s.shape == g.shape, (name, s.shape, g.shape)
    err = (s - g).abs().max() / (g.abs().max() + 1e-30)
    assert err <= 1e-4, (name, float(err))


@pytest.mark.parametrize("case,T,tasks,nq", [("mini_T16_all", 16, ALL, 8),
                                             ("mini_T32_stitch", 32, ["depth", "flow_2d_backward", "dyn_mask", "track_2d"], 12)])
def test_oracle_matches_reference_goldens(mini, case, T, tasks, nq):
    cfg, sd = mini
    gold = np.load(os.path.join(GOLD, case + ".npz"))
    batch = make_batch(T, nq)
    trace = []
    with torch.no_grad():
        out = OracleModel(sd, cfg, use_intrinsics=True).forward(batch, tasks, trace=trace)
        feats = encoder_forward(sd, batch["rgb_b3thw"][:, :, :16], cfg)
    for li in sorted(set([0, 1, cfg.depth] + list(cfg.hooks))):
        _cmp(f"feat{li}", feats[li], gold[f"feat{li}"])
    keys = [k for k in gold.files if not k.startswith("feat") and not k.startswith("trace")]
    assert sorted(keys) == sorted(out.keys())
    for k in keys:
        _cmp(k, out[k], gold[k])
    nwin = len([k for k in gold.files if k.endswith("_labels") and "prompt" not in k])
    assert nwin == len(trace) == (T - 16) // 8 + 1
    for w in range(nwin):
        assert np.array_equal(gold[f"trace{w}_labels"], trace[w]["labels"].float().numpy()), w
        assert np.array_equal(gold[f"trace{w}_prompt_labels"], trace[w]["prompt_labels"].float().numpy()), w
        assert np.array_equal(gold[f"trace{w}_queries"][:, 0], trace[w]["queries"][:, 0].numpy()), w
        _cmp(f"trace{w}_queries", trace[w]["queries"], gold[f"trace{w}_queries"])


def test_full_size_goldens_are_present_and_pinned():
    """The full-size (1408 x 40) fixtures were produced by the reference and checked against the oracle at
    generation time (oracle_vs_reference_full.json); re-running them on CPU takes minutes, so here only the
    recorded agreement is asserted.  The GPU suite compares the engine with these vectors."""
    import json

    rep = json.load(open(os.path.join(GOLD, "oracle_vs_reference_full.json")))["full_T16_all"]
    assert max(v["oracle_rel_err"] for v in rep["tensors"].values()) <= 1e-4
    assert rep.get("tracker_state_windows", 0) >= 1
    g = np.load(os.path.join(GOLD, "full_T16_all.npz"))
    assert "depth_est_b1thw" in g.files and "feat36" in g.files


def test_joint_oracle_matches_reference_flow_golden(mini):
    """Rows a11 / f1: the 3-window joint depth + camera flow.  tests/golden/mini_T32_joint.npz was produced by the reference's
    own joint_windowed_estimation with its two random draws replaced by the fixed stand-ins of oracle/joint_oracle.py
    (tools/gen_golden_joint.py); the oracle must reproduce the stitched outputs, the per-seam thresholds (exact q98) and
    the per-seam similarity transforms."""
    import json

    cfg, sd = mini
    gold = np.load(os.path.join(GOLD, "mini_T32_joint.npz"))
    batch = make_batch(32, 4)
    om = OracleModel(sd, cfg, use_intrinsics=True, seam="fixed")
    with torch.no_grad():
        out = om.forward(batch, ["depth", "camray"])
    for k in ("depth_est_b1thw", "traj3d_est_b16t", "traj3d_intrinsics_est_b16t"):
        _cmp(k, out[k], gold[k])
    assert len(om.seam_log) == 2
    for i, s in enumerate(om.seam_log):
        assert abs(s["thr"] - float(gold[f"seam{i}_thr"])) <= 1e-6 * abs(s["thr"])
        assert np.abs(s["T"] - gold[f"seam{i}_T"]).max() <= 1e-5 * np.abs(gold[f"seam{i}_T"]).max()
        assert abs(s["s"] - float(gold[f"seam{i}_s"])) <= 1e-6 * s["s"]
    rep = json.load(open(os.path.join(GOLD, "oracle_vs_reference_joint.json")))
    assert max(rep.values()) <= 1e-4  # function-level pins (generate_point_map, apply) recorded at generation time


def test_default_config_oracle_matches_reference_flow_golden(mini):
    """Rows a10 / f1, the shipped default (use_intrinsics=false, fixed_intrinsics=true): tests/golden/mini_T32_default_config.npz
    was produced by the reference's own forward — K estimated on the first window by rays_to_cameras_and_fixed_per_frame_
    intrinsics (geometry_utils.py:493-579), reported for the later windows whose rotations use the input K
    (dense_heads.py:303-334), joint alignment on top — with the two cv2 calls replaced by the deterministic stand-ins of
    oracle/l4p_oracle.py installed in its cv2 stub (tools/gen_golden_intrinsics.py).  The oracle must reproduce it."""
    import json

    cfg, sd = mini
    gold = np.load(os.path.join(GOLD, "mini_T32_default_config.npz"))
    om = OracleModel(sd, cfg, use_intrinsics=False, seam="fixed")
    with torch.no_grad():
        out = om.forward(make_batch(32, 4), ["depth", "camray"])
    for k in ("depth_est_b1thw", "traj3d_est_b16t", "traj3d_intrinsics_est_b16t"):
        _cmp(k, out[k], gold[k])
    K = out["traj3d_intrinsics_est_b16t"]
    assert float((K - K[:, :, :1]).abs().max()) == 0.0  # one estimate for the whole clip
    rep = json.load(open(os.path.join(GOLD, "oracle_vs_reference_intrinsics.json")))
    assert max(v for k, v in rep.items() if not k.startswith("synthetic_K_recovery")) <= 1e-4
    assert max(v for k, v in rep.items() if k.startswith("synthetic_K_recovery")) <= 2e-2


def test_k_estimation_stand_ins_and_supplied_estimate():
    """The deterministic stand-ins for cv2.findHomography / cv2.RQDecomp3x3 recover a known homography / factorisation, and
    supplying a K (k_override) is the same as estimating that K."""
    from oracle import l4p_oracle as lo

    g = np.random.default_rng(3)
    Ht = np.array([[1.1, 0.05, 0.3], [-0.02, 0.9, -0.2], [0.01, 0.02, 1.0]])
    src = g.normal(size=(200, 2))
    d = (Ht @ np.c_[src, np.ones(200)].T).T
    H, _ = lo.dlt_homography(src, d[:, :2] / d[:, 2:])
    assert np.abs(H - Ht).max() <= 1e-9
    M = g.normal(size=(3, 3))
    if np.linalg.det(M) < 0:
        M = -M
    _, K, R = lo.rq3(M)
    assert np.abs(K @ R - M).max() <= 1e-12 and np.abs(np.tril(K, -1)).max() <= 1e-12 and (np.diag(K) > 0).all()
    assert np.abs(R @ R.T - np.eye(3)).max() <= 1e-12 and np.linalg.det(R) > 0


def test_joint_oracle_pieces():
    from oracle import joint_oracle as jo

    # Umeyama recovers a known similarity exactly, incl. a reflection-prone (planar-ish) configuration
    g = np.random.default_rng(0)
    src = g.normal(size=(200, 3))
    src[:, 2] *= 1e-3
    A = np.linalg.qr(g.normal(size=(3, 3)))[0]
    R = A if np.linalg.det(A) > 0 else -A
    dst = 2.5 * src @ R.T + np.array([0.3, -1.0, 4.0])
    rel = jo.umeyama(src, dst)
    assert abs(rel["s"] - 2.5) < 1e-9 and np.abs(rel["T"][:3, :3] / rel["s"] - R).max() < 1e-6
    # the fixed permutation is a permutation; the engine's pixel subset takes one pixel per stride cell
    for n in (150528, 1000, 7919 * 2):
        assert np.array_equal(np.sort(jo.fixed_permutation(n)), np.arange(n))
    sub = jo.engine_pixel_subset(224, 224, 10)
    assert np.array_equal(sub // 10, np.arange(224 * 224 // 10))
    # hash_u32 known answers (uint32 arithmetic of csrc/umeyama.hip:hash_u32, computed by hand in Python ints)
    def h(x):
        x = (x * 747796405 + 2891336453) & 0xFFFFFFFF
        w = (((x >> ((x >> 28) + 4)) ^ x) * 277803737) & 0xFFFFFFFF
        return (w >> 22) ^ w
    for x in (0, 1, 12345, 0xFFFFFFFF, jo.ENGINE_SEED):
        assert int(jo.hash_u32(np.array([x]))[0]) == h(x)
    # engine_ransac finds the similarity among 30 % gross outliers
    dst2 = dst.copy()
    dst2[:60] += g.normal(size=(60, 3)) * 5
    rel2, inl = jo.engine_ransac(src.astype(np.float32), dst2.astype(np.float32), thr=1e-3)
    assert inl.sum() >= 135 and abs(rel2["s"] - 2.5) < 1e-3


def test_oracle_single_window_entry_matches_reference_golden(mini):
    """always_use_windowed_version=False, T == 16: forward_single_window (l4p_videomae.py:234-254); the tracker's plain
    forward (sparse_heads.py:497-600) uses the raw last feature, the caller's labels and returns unmasked outputs."""
    from tests.golden_utils import single_window_batch

    cfg, sd = mini
    gold = np.load(os.path.join(GOLD, "mini_T16_single_window.npz"))
    om = OracleModel(sd, cfg, use_intrinsics=True)
    om.always_use_windowed_version = False
    with torch.no_grad():
        out = om.forward(single_window_batch(), ["track_2d", "depth", "flow_2d_backward"])
    assert sorted(out.keys()) == sorted(gold.files)
    for k in gold.files:
        _cmp(k, out[k], gold[k])


def test_full_size_joint_and_q64_goldens_are_pinned():
    """tests/golden/full_T40_joint.npz (4 windows / 3 seams, all tasks) and full_T16_q64.npz (the benchmark's 64 queries) come
    from the imported reference at the full geometry (tools/gen_golden_full_joint.py); the oracle's agreement over the FULL
    tensors was asserted when they were generated and is recorded — re-running 4 full-size windows on the CPU takes minutes."""
    import json

    rep = json.load(open(os.path.join(GOLD, "oracle_vs_reference_full_joint.json")))
    errs = {k: v for k, v in rep.items() if k.endswith("_oracle_rel_err")}
    assert len(errs) == 11 and max(errs.values()) <= 1e-4, errs
    g = np.load(os.path.join(GOLD, "full_T40_joint.npz"))
    for k in ("depth_est_b1thw", "traj3d_est_b16t", "engine.depth_est_b1thw", "engine.traj3d_est_b16t", "seam2_T", "engine.seam2_T"):
        assert k in g.files, k
    # (name-seeded random weights: neighbouring windows' depth / pose estimates are mutually inconsistent, so at the reference's
    #  threshold of 1 % of the q98 depth only a handful of the 15051 sampled points agree — the seams are still well defined)
    assert all(rep[f"T40_engine_seam{i}"]["inliers"] >= 1 and rep[f"T40_engine_seam{i}"]["n"] == 15051 for i in range(3)), rep
    q = np.load(os.path.join(GOLD, "full_T16_q64.npz"))
    assert q["track_2d_traj_est_bn2t"].shape == (1, 64, 2, 16)


def test_variable_intrinsics_oracle_matches_reference_fixture():
    """fixed_intrinsics=False (geometry_utils.py:582-654): tests/golden/intrinsics_variable.npz holds the reference's own
    rays_to_cameras_and_variable_per_frame_intrinsics on seeded synthetic ray maps with its two cv2 calls replaced by the oracle's
    deterministic stand-ins (tools/gen_golden_intrinsics.py); the oracle's restatement must reproduce it."""
    from oracle import l4p_oracle as lo
    from tests.golden_utils import synthetic_rays

    gold = np.load(os.path.join(GOLD, "intrinsics_variable.npz"))
    rays, Ks = synthetic_rays()
    E, K = lo.rays_to_cameras_variable_intrinsics(rays, (224, 224))
    assert np.abs(E.numpy() - gold["E"]).max() <= 1e-5 * np.abs(gold["E"]).max()
    assert np.abs(K.numpy() - gold["K"]).max() <= 1e-5 * np.abs(gold["K"]).max()
    # every frame recovers its item's camera (one K per batch item in the synthetic data, estimated per frame)
    K_ray = lo.denormalize_intrinsics(lo.normalize_intrinsics(K, 224, 224), 16, 16)
    for b, Kb in enumerate(Ks):
        for t in range(K.shape[-1]):
            assert float((K_ray[b, :3, :3, t] - Kb).abs().max() / Kb.abs().max()) <= 2e-2, (b, t)


def test_reference_autocast_drift_and_full_size_traces_are_well_formed():
    """tools/gen_golden_full_autocast.py: the reference's own bf16-autocast drift (what the bf16 engine's gates are) and the
    integer / boolean tracker state of the full-size multi-window goldens.  Checked here without a GPU: every case ran (no
    autocast failure), the figures are in the range SURVEY.md §7 measured (1.35e-2 on the encoder), and the recorded state is
    self-consistent — the validity mask is the function of the window's query times that sparse_heads.py:306-319 states, labels
    follow it (:326-335), prompt labels switch on after the first window a track was valid in (:389-393), re-seeded query times
    only move forward (:455-486)."""
    import json

    with open(os.path.join(GOLD, "reference_autocast_drift.json")) as f:
        rep = json.load(f)
    for case in ("full_T16_all", "full_T24_windows", "full_T40_track24", "full_T16_q64", "mini_T16_all", "mini_T32_stitch",
                 "mini_T16_single_window"):
        assert case in rep and "autocast_failed" not in rep[case], case
    assert 1.0e-2 <= rep["full_T16_all"]["feat40"]["rel_l2"] <= 1.7e-2
    assert rep["full_T16_all"]["traj3d_intrinsics_est_b16t"]["rel_l2"] == 0.0  # use_intrinsics=True: the input K passes through
    assert rep["full_T40_track24"]["tracks"] == 24 and 0 <= rep["full_T40_track24"]["tracks_with_differing_integer_state"] <= 24
    for name, nwin, n in (("full_T24_windows", 2, 4), ("full_T40_joint", 4, 8), ("full_T40_track24", 4, 24)):
        g = np.load(os.path.join(GOLD, name + ".npz"))
        prev_t = None
        was_valid = np.zeros(n, dtype=bool)
        for w in range(nwin):
            lab, pl, q, vt = (g[f"trace{w}_{k}"] for k in ("labels", "prompt_labels", "queries", "valid_t"))
            assert lab.shape == (n,) and pl.shape == (n,) and q.shape == (n, 3) and vt.shape == (n, 16), (name, w)
            want = (np.arange(16)[None, :] + 0.5 - q[:, 0:1]) >= 0  # query times are stored relative to the window start
            assert np.array_equal(vt, want), (name, w)
            valid_n = vt.any(-1)
            # (a query still equal to the caller's input keeps label 1 even before its start frame, :330-332)
            assert set(np.unique(lab)) <= {0.0, 1.0, 2.0} and not (valid_n & (lab == 0)).any() and not (~valid_n & (lab == 2)).any(), (name, w)
            assert np.array_equal(pl == 1, was_valid), (name, w)
            was_valid |= valid_n
            t_abs = q[:, 0] + 8 * w
            if prev_t is not None:
                assert (t_abs >= prev_t).all(), (name, w)
            prev_t = t_abs
            if w < nwin - 1:
                b = g[f"trace{w}_best_vis_id"]
                assert b.shape == (n,) and b.min() >= 0 and b.max() < 8, (name, w)
