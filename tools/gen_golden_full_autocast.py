"""Reference-anchored numbers for the BENCHMARKED (bf16) engine, at the full geometry, from the REAL reference (runs only where
/root/reference exists; round-3 review, "next round" item 1):

A. tests/golden/reference_autocast_drift.json — the reference's OWN mixed-precision drift: the imported reference run under
   ``torch.autocast("cpu", dtype=torch.bfloat16)`` (its demo runs "16-mixed", /root/reference/demo/demo.py:22-23,35-43) against
   its own fp32 run on the very inputs the goldens use (full_T16_all: 16 frames, all five tasks, 8 tracks; full_T24_windows:
   24 frames = 2 windows, depth / flow / mask / 4 tracks; T40: 4 windows, 24 tracks).  Per output and per tapped feature: rel-L2
   and max-relative-to-max over the FULL tensors; for the tracker the number of tracks whose integer / boolean window state
   (labels, prompt labels, re-seeded query times) differs between the two runs.  The engine's bf16 gates in
   tests/test_full_model_gpu.py are "engine drift <= reference autocast drift" on these figures.
B. integer / boolean tracker state at the FULL geometry over several windows: ``trace{w}_labels / _prompt_labels / _queries``
   recorded from the reference (as tools/gen_golden.py does at the mini geometry) are ADDED to full_T24_windows.npz (2 windows)
   and full_T40_joint.npz (4 windows, 8 tracks) after checking that this run reproduces the tracks those files hold; the
   oracle's trace on the reference's features equals the reference's (asserted) and contributes ``valid_t`` and ``best_vis_id``.
C. tests/golden/full_T40_track24.npz — 24 tracks with mixed start frames over 4 windows (complete outputs + trace): a sample
   large enough for a count-bounded statement on the bf16 engine.

  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_full_autocast.py        (~20 minutes on 8 cores)
Only data is written."""
from __future__ import annotations

import json
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import grid_queries, make_batch, sample_indices, single_window_batch
from tools.gen_golden import build_reference, install_stubs

# --dtype float16 (round 5): the drift of the mode the reference's demo actually ships - Fabric "16-mixed" is float16 autocast
# (/root/reference/demo/demo.py:22-23, l4p/models/utils.py:57-58) - into reference_autocast_drift_f16.json; the golden .npz files
# (parts B, C: fp32 runs) are only rewritten by the default bfloat16 run.
AC_DTYPE = torch.float16 if "--dtype" in sys.argv and sys.argv[sys.argv.index("--dtype") + 1] == "float16" else torch.bfloat16
WRITE_GOLD = AC_DTYPE == torch.bfloat16
ALL = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]
TRACK = ["track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t"]
GOLD = os.path.join(ROOT, "tests", "golden")


def run_ref(model, batch, tasks, autocast: bool):
    """-> (outputs, per-window trace, enc_features_bpc_2dlist).  forward_windowed_core calls self.forward(...) directly (no module
    hooks fire), so the bound method is shadowed to record the integer state each window starts from."""
    trace = []
    head = model.task_heads["track_2d"]
    orig_forward = head.forward

    def spy_forward(*a, **kwargs):
        if "track_2d_promptfeaturelabels_bn" in kwargs:
            trace.append({"labels": kwargs["track_2d_pointlabels_bn"][0].clone().float(),
                          "queries": kwargs["track_2d_pointquerries_bn3"][0].clone().float(),
                          "prompt_labels": kwargs["track_2d_promptfeaturelabels_bn"][0].clone().float()})
        return orig_forward(*a, **kwargs)

    head.forward = spy_forward
    try:
        with torch.no_grad(), torch.autocast("cpu", dtype=AC_DTYPE, enabled=autocast):
            out = model.forward({k: v.clone() for k, v in batch.items()}, list(tasks))
    finally:
        del head.forward
    feats = out.pop("enc_features_bpc_2dlist")
    return {k: v for k, v in out.items() if torch.is_tensor(v)}, trace, feats


def drift(a: torch.Tensor, b: torch.Tensor) -> dict:
    """a: autocast run, b: fp32 run (full tensors)."""
    a, b = a.detach().float(), b.detach().float()
    return {"rel_l2": float((a - b).norm() / (b.norm() + 1e-30)), "max_rel": float((a - b).abs().max() / (b.abs().max() + 1e-30))}


def differing_tracks(ta, tb) -> int:
    n = ta[0]["labels"].numel()
    d = torch.zeros(n, dtype=torch.bool)
    for a, b in zip(ta, tb):
        d |= a["labels"] != b["labels"]
        d |= a["prompt_labels"] != b["prompt_labels"]
        d |= a["queries"][:, 0] != b["queries"][:, 0]
    return int(d.sum())


def case(model, name, T, nq, tasks, taps, report, batch=None, first=None):
    """``first``: also report the tracker drift restricted to the first ``first`` queries (tracks are independent)."""
    batch = make_batch(T, nq) if batch is None else batch
    t0 = time.time()
    o32, t32, f32 = run_ref(model, batch, tasks, False)
    t1 = time.time()
    rep = {"fp32_seconds": round(t1 - t0, 1)}
    try:
        o16, t16, f16 = run_ref(model, batch, tasks, True)
        rep["autocast_seconds"] = round(time.time() - t1, 1)
        for li in taps:
            rep[f"feat{li}"] = drift(f16[0][li], f32[0][li])
        for k in o32:
            rep[k] = drift(o16[k], o32[k])
            if first is not None and k in TRACK:
                rep[f"{k}[:{first}]"] = drift(o16[k][:, :first], o32[k][:, :first])
        if t32:
            rep["tracks"] = int(t32[0]["labels"].numel())
            rep["tracks_with_differing_integer_state"] = differing_tracks(t16, t32)
            if first is not None:
                sub = [[{k: v[:first] for k, v in w.items()} for w in tr] for tr in (t16, t32)]
                rep[f"tracks_with_differing_integer_state[:{first}]"] = differing_tracks(*sub)
        del o16, f16
    except Exception as e:  # an op the reference leaves outside autocast's cast lists may refuse bf16 on CPU
        rep["autocast_failed"] = f"{type(e).__name__}: {e}"[:400]
    report[name] = rep
    print(name, json.dumps(rep, indent=1), flush=True)
    return batch, o32, t32, f32


def oracle_trace(sd, cfg, feats2d, batch, strides, ref_trace):
    """The oracle's tracker on the REFERENCE's last-layer features: its integer state must equal the reference's; it adds the
    validity masks and the argmax index the reference does not expose."""
    from oracle import l4p_oracle as lo

    tr = []
    with torch.no_grad():
        oo = lo.track_windowed(sd, cfg, [f[-1].float() for f in feats2d], batch["track_2d_pointquerries_bn3"],
                               batch["track_2d_pointlabels_bn"], strides, trace=tr)
    assert len(tr) == len(ref_trace)
    for w, (a, b) in enumerate(zip(ref_trace, tr)):
        assert torch.equal(a["labels"], b["labels"].float()), (w, "labels")
        assert torch.equal(a["prompt_labels"], b["prompt_labels"].float()), (w, "prompt_labels")
        assert torch.equal(a["queries"][:, 0], b["queries"][:, 0]), (w, "query times")
    return oo, tr


def trace_arrays(ref_trace, or_trace, sel=slice(None)):
    npz = {}
    for w, (a, b) in enumerate(zip(ref_trace, or_trace)):
        npz[f"trace{w}_labels"] = a["labels"][sel].numpy()
        npz[f"trace{w}_prompt_labels"] = a["prompt_labels"][sel].numpy()
        npz[f"trace{w}_queries"] = a["queries"][sel].numpy()
        npz[f"trace{w}_valid_t"] = b["valid_t"][sel].numpy()
        if "best_vis_id" in b:
            npz[f"trace{w}_best_vis_id"] = b["best_vis_id"][sel].numpy().astype(np.int64)
    return npz


def main():
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = ModelCfg.full()
    t0 = time.time()
    model = build_reference(cfg)
    sd = seeded_state_dict(cfg)
    model.load_state_dict(sd, strict=True)
    print(f"built + loaded in {time.time() - t0:.1f}s", flush=True)
    report = {"what": f"reference under torch.autocast('cpu', {AC_DTYPE}) vs its own fp32 run, full geometry, full tensors",
              "torch": torch.__version__}
    taps = [14, 21, 28, 36, 40]

    # ---- T16, all tasks (the inputs of full_T16_all.npz) ---------------------------------------------------------------------
    batch, o32, t32, f32 = case(model, "full_T16_all", 16, 8, ALL, taps, report)
    g = np.load(os.path.join(GOLD, "full_T16_all.npz"))
    for k in o32:  # this run is the run that file was sampled from
        v = o32[k].float().reshape(-1)
        s = v[sample_indices(v.numel())] if v.numel() > 4096 else v
        assert np.abs(s.numpy() - g[k].reshape(-1)).max() <= 1e-5 * np.abs(g[k]).max(), k
    del o32, f32

    # ---- T24, two windows (the inputs of full_T24_windows.npz) + trace ------------------------------------------------------
    tasks24 = ["depth", "flow_2d_backward", "dyn_mask", "track_2d"]
    batch, o32, t32, f32 = case(model, "full_T24_windows", 24, 4, tasks24, taps, report)
    path = os.path.join(GOLD, "full_T24_windows.npz")
    g = dict(np.load(path))
    for k in o32:
        v = o32[k].float().reshape(-1)
        s = v[sample_indices(v.numel())] if v.numel() > 4096 else v
        assert np.abs(s.numpy() - g[k].reshape(-1)).max() <= 1e-5 * np.abs(g[k]).max(), k
    _, otr = oracle_trace(sd, cfg, f32, batch, [0, 8], t32)
    g = {k: v for k, v in g.items() if not k.startswith("trace")}
    g.update(trace_arrays(t32, otr))
    if WRITE_GOLD:
        np.savez_compressed(path, **g)
    print("full_T24_windows.npz: trace arrays of", len(t32), "windows added", flush=True)
    del o32, f32

    # ---- T40, four windows, 24 tracks (tracker only: it reads the last-layer features alone) ---------------------------------
    batch, o32, t32, f32 = case(model, "full_T40_track24", 40, 24, ["track_2d"], [40], report, first=8)
    strides = [0, 8, 16, 24]
    oo, otr = oracle_trace(sd, cfg, f32, batch, strides, t32)
    for k in TRACK:
        e = float((oo[k] - o32[k]).abs().max() / o32[k].abs().max())
        assert e <= 1e-4, (k, e)
    npz = {k: o32[k].float().numpy() for k in TRACK}
    npz.update(trace_arrays(t32, otr))
    if WRITE_GOLD:
        np.savez_compressed(os.path.join(GOLD, "full_T40_track24.npz"), **npz)
    # the first 8 of these queries are full_T40_joint.npz's (make_batch draws query i from i alone): same tracks, same state
    path = os.path.join(GOLD, "full_T40_joint.npz")
    g = dict(np.load(path))
    for k in TRACK:
        e = np.abs(o32[k][:, :8].float().numpy() - g[k]).max() / np.abs(g[k]).max()
        print("T40 first 8 tracks vs full_T40_joint.npz", k, f"{e:.2e}", flush=True)
        assert e <= 1e-4, (k, e)
    g = {k: v for k, v in g.items() if not k.startswith("trace")}
    g.update(trace_arrays(t32, otr, slice(0, 8)))
    if WRITE_GOLD:
        np.savez_compressed(path, **g)
    print("full_T40_joint.npz: trace arrays of", len(t32), "windows added", flush=True)

    del o32, f32

    # ---- the benchmark's own query set: 64 grid queries on the golden clip (the inputs of full_T16_q64.npz) -------------------
    b16 = make_batch(16, 1)
    b16["track_2d_pointquerries_bn3"] = grid_queries(64)
    b16["track_2d_pointlabels_bn"] = torch.ones(1, 64)
    _, o32, _, _ = case(model, "full_T16_q64", 16, 64, ["track_2d"], [40], report, batch=b16)
    g = np.load(os.path.join(GOLD, "full_T16_q64.npz"))
    for k in TRACK:
        assert np.abs(o32[k].float().numpy() - g[k]).max() <= 1e-5 * np.abs(g[k]).max(), k
    del model, o32

    # ---- the mini geometry (tests/test_track_gpu.py, test_encoder_dpt_gpu.py gate their bf16 engine on these) ------------------
    mcfg = ModelCfg.mini()
    mini = build_reference(mcfg)
    mini.load_state_dict(seeded_state_dict(mcfg), strict=True)
    mtaps = sorted(set(list(mcfg.hooks) + [mcfg.depth]))
    case(mini, "mini_T16_all", 16, 8, ALL, mtaps, report)
    case(mini, "mini_T32_stitch", 32, 12, ["depth", "flow_2d_backward", "dyn_mask", "track_2d"], mtaps, report)
    case(mini, "mini_T40_track9", 40, 9, ["track_2d"], [mcfg.depth], report)
    mini.always_use_windowed_version = False
    try:
        with torch.no_grad():
            bs = single_window_batch()
            tasks = ["track_2d", "depth", "flow_2d_backward"]
            o32 = mini.forward({k: v.clone() for k, v in bs.items()}, tasks)
            with torch.autocast("cpu", dtype=AC_DTYPE):
                o16 = mini.forward({k: v.clone() for k, v in bs.items()}, tasks)
        report["mini_T16_single_window"] = {k: drift(o16[k], o32[k]) for k in o32 if torch.is_tensor(o32[k])}
        print("mini_T16_single_window", json.dumps(report["mini_T16_single_window"], indent=1), flush=True)
    finally:
        mini.always_use_windowed_version = True

    with open(os.path.join(GOLD, "reference_autocast_drift.json" if WRITE_GOLD else "reference_autocast_drift_f16.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
