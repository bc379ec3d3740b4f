"""Generate tests/golden/* from the REAL reference (runs only where /root/reference exists).

  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--full]

What it does
  1. imports NVlabs/L4P from /root/reference with three import stubs (timm, skimage, cv2 are not
     installed; none of the stubbed functions is on the paths exercised here),
  2. builds the reference model (mini: 704-wide / 4-deep encoder, and with --full the shipped
     VideoMAE-v2-giant geometry), loads OUR name-seeded weights strictly (which also proves that
     l4p_amd.weights.state_dict_schema matches the reference key set and shapes),
  3. runs seeded synthetic inputs through the reference and writes small fixtures: the state-dict
     manifest, sampled output values + statistics, the tracker's integer/boolean window state,
  4. runs the oracle (oracle/l4p_oracle.py) on the same inputs and records max |oracle - reference|
     over the FULL tensors in tests/golden/oracle_vs_reference_*.json (asserting <= 1e-4 relative).

Only data (inputs are regenerated from seeds; outputs are sampled values) is committed — no
reference source travels.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np
import torch

from l4p_amd.weights import ModelCfg, seeded_state_dict, state_dict_schema
from tests.golden_utils import make_batch, sample_indices


def install_stubs():
    def ident(x=None, *a, **k):
        return x

    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")
    tl.drop_path = lambda x, p=0.0, training=False: x
    tl.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    tr = types.ModuleType("timm.models.registry")
    tr.register_model = ident
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl, "timm.models.registry": tr})
    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.measure")
    skm.ransac = None
    skt = types.ModuleType("skimage.transform")
    skt.SimilarityTransform = None
    sys.modules.update({"skimage": sk, "skimage.measure": skm, "skimage.transform": skt})
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, REF)


def build_reference(cfg: ModelCfg):
    """The reference L4P_VideoMAE with the five heads of configs/model.yaml (use_intrinsics=True: the
    cv2 path cannot run here).  For the mini geometry the giant encoder hard-coded in
    L4P_VideoMAE.__init__ is swapped for a small VideoMAEEncoder built from the same class."""
    from functools import partial

    from l4p.models.l4p_videomae import L4P_VideoMAE, VideoMAEEncoder
    from l4p.models.task_heads import dense_heads as dh
    from l4p.models.task_heads import sparse_heads as sh

    hooks = list(cfg.hooks)
    kw = dict(depth=cfg.depth, embed_dim=cfg.dim)
    heads = torch.nn.ModuleDict(dict(
        flow_2d_backward=dh.VideoMAEFlowDPTHead(task_name="flow_2d_backward", out_nchan=2, hooks_idx=hooks, **kw),
        depth=dh.VideoMAEDepthDPTHead(task_name="depth", out_nchan=1, depth_fn="exp", hooks_idx=hooks,
                                      align_window_overlap_fn="inverse", **kw),
        camray=dh.VideoMAETraj3DDPTHead(task_name="traj3d", hooks_idx=hooks, use_intrinsics=True,
                                        fixed_intrinsics=True, **kw),
        dyn_mask=dh.VideoMAEDynMaskDPTHead(task_name="dyn_mask", out_nchan=1, apply_fn="linear", hooks_idx=hooks, **kw),
        track_2d=sh.VideoMAETrack2DSamHead(task_name="track_2d", prompt_embed_dim=cfg.dim, estimate_vis=True,
                                           estimate_depth=True, sam_head_depth=2, num_point_embeddings=2,
                                           prompt_using_features=True, attend_to_past=True,
                                           modify_pointlabels_for_windowing=True, estimation_directions=[1],
                                           depth_fn="exp", vis_fn="linear"),
    ))
    if cfg.dim == 1408 and cfg.depth == 40:
        model = L4P_VideoMAE(task_heads=heads, always_use_windowed_version=True, joint_alignment=True)
    else:
        model = L4P_VideoMAE.__new__(L4P_VideoMAE)
        torch.nn.Module.__init__(model)
        model.video_encoder = VideoMAEEncoder(
            img_size=cfg.img, patch_size=cfg.patch[1], in_chans=3, num_classes=0, embed_dim=cfg.dim, depth=cfg.depth,
            num_heads=cfg.heads, mlp_ratio=48 / 11, qkv_bias=True, qk_scale=None, drop_rate=0, attn_drop_rate=0,
            drop_path_rate=0, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), init_values=0.0, tubelet_size=2,
            use_learnable_pos_emb=False, with_cp=False, all_frames=16, cos_attn=False)
        model.task_heads = heads
        model.window_size = (16, 224, 224)
        model.window_stride_T = 8
        model.always_use_windowed_version = True
        model.joint_alignment = True
    return model.eval()


def summarize(t: torch.Tensor, n: int = 4096):
    t = t.detach().float().reshape(-1)
    idx = sample_indices(t.numel(), n)
    return {"shape": None, "idx_n": int(idx.numel()), "vals": t[idx].numpy().astype(np.float32),
            "mean": float(t.mean()), "std": float(t.std()) if t.numel() > 1 else 0.0, "absmax": float(t.abs().max())}


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


def run_case(name: str, cfg: ModelCfg, model, sd, T: int, tasks, nq: int, out_dir: str, report: dict):
    from oracle.l4p_oracle import OracleModel

    batch = make_batch(T, nq)
    trace_ref = []

    # forward_windowed_core calls self.forward(...) directly (no module hooks fire), so shadow the bound method
    head = model.task_heads["track_2d"]
    orig_forward = head.forward

    def spy_forward(*a, **kwargs):
        if "track_2d_promptfeaturelabels_bn" in kwargs:  # the per-window call
            trace_ref.append({
                "labels": kwargs["track_2d_pointlabels_bn"][0].clone(),
                "queries": kwargs["track_2d_pointquerries_bn3"][0].clone(),
                "prompt_labels": kwargs["track_2d_promptfeaturelabels_bn"][0].clone(),
            })
        return orig_forward(*a, **kwargs)

    head.forward = spy_forward
    t0 = time.time()
    with torch.no_grad():
        out = model.forward({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}, list(tasks))
    t_ref = time.time() - t0
    del head.forward
    feats2d = out.pop("enc_features_bpc_2dlist")

    trace_or = []
    t0 = time.time()
    with torch.no_grad():
        oout = OracleModel(sd, cfg, use_intrinsics=True).forward(batch, list(tasks), trace=trace_or)
        from oracle.l4p_oracle import encoder_forward
        ofeats = encoder_forward(sd, batch["rgb_b3thw"][:, :, :16], cfg)
    t_or = time.time() - t0

    npz = {}
    rep = {"ref_seconds": round(t_ref, 2), "oracle_seconds": round(t_or, 2), "tensors": {}}
    for li in sorted(set([0, 1, cfg.depth] + list(cfg.hooks))):
        s = summarize(feats2d[0][li])
        npz[f"feat{li}"] = s["vals"]
        e = rel_err(ofeats[li], feats2d[0][li])
        rep["tensors"][f"feat{li}"] = {"shape": list(feats2d[0][li].shape), "mean": s["mean"], "std": s["std"],
                                       "absmax": s["absmax"], "oracle_rel_err": e}
        assert e <= 1e-4, (name, li, e)
    for k, v in out.items():
        s = summarize(v)
        npz[k] = s["vals"] if v.numel() > 4096 else v.detach().float().numpy()
        e = rel_err(oout[k], v)
        rep["tensors"][k] = {"shape": list(v.shape), "mean": s["mean"], "std": s["std"], "absmax": s["absmax"],
                             "oracle_rel_err": e}
        assert e <= 1e-4, (name, k, e)
    assert set(oout.keys()) == set(out.keys()), (sorted(oout.keys()), sorted(out.keys()))
    # integer / boolean tracker state, bit-exact between oracle and reference
    if trace_ref:
        assert len(trace_ref) == len(trace_or)
        for w, (a, b) in enumerate(zip(trace_ref, trace_or)):
            assert torch.equal(a["labels"].float(), b["labels"].float()), (name, w, "labels")
            assert torch.equal(a["prompt_labels"].float(), b["prompt_labels"].float()), (name, w, "prompt_labels")
            assert torch.equal(a["queries"][:, 0], b["queries"][:, 0]), (name, w, "query times")
            npz[f"trace{w}_labels"] = a["labels"].float().numpy()
            npz[f"trace{w}_prompt_labels"] = a["prompt_labels"].float().numpy()
            npz[f"trace{w}_queries"] = a["queries"].float().numpy()
        rep["tracker_state_windows"] = len(trace_ref)
    np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **npz)
    report[name] = rep
    print(f"[{name}] reference {t_ref:.1f}s oracle {t_or:.1f}s  max oracle_rel_err "
          f"{max(v['oracle_rel_err'] for v in rep['tensors'].values()):.2e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also generate the full-size (1408x40) fixtures (minutes)")
    args = ap.parse_args()
    install_stubs()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 8)

    cfgs = [("mini", ModelCfg.mini())] + ([("full", ModelCfg.full())] if args.full else [])
    for cname, cfg in cfgs:
        report = {}
        t0 = time.time()
        model = build_reference(cfg)
        ref_sd = model.state_dict()
        manifest = {k: list(v.shape) for k, v in ref_sd.items()}
        with open(os.path.join(out_dir, f"manifest_{cname}.json"), "w") as f:
            json.dump(manifest, f, indent=0, sort_keys=True)
        schema = state_dict_schema(cfg)
        assert set(schema) == set(manifest), (sorted(set(schema) ^ set(manifest))[:10])
        for k in schema:
            assert list(schema[k]) == manifest[k], (k, schema[k], manifest[k])
        sd = seeded_state_dict(cfg)
        model.load_state_dict(sd, strict=True)
        print(f"[{cname}] built + loaded {len(sd)} tensors in {time.time() - t0:.1f}s")
        all_tasks = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]
        if cname == "mini":
            run_case("mini_T16_all", cfg, model, sd, 16, all_tasks, 8, out_dir, report)
            run_case("mini_T32_stitch", cfg, model, sd, 32, ["depth", "flow_2d_backward", "dyn_mask", "track_2d"], 12,
                     out_dir, report)
        else:
            run_case("full_T16_all", cfg, model, sd, 16, all_tasks, 8, out_dir, report)
        with open(os.path.join(out_dir, f"oracle_vs_reference_{cname}.json"), "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
        del model


if __name__ == "__main__":
    main()
