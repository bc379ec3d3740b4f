"""bench.py's algorithmic FLOP model (what `roofline.achieved` is computed from) against what the engine EXECUTES:
the sum of 2*M*N*K over the per-shape event profile of a c3 step (profiles/r04_c3_per_shape_event_profile.txt, written by
tools/prof_detail.py on the GPU), minus the padding the kernels add (head dim 88 -> 96 in the QKV projection, per-tap
channels 176 -> 192 in the tracker's last up-scaling, patch vector 1176 -> 1216, and in the tracker's folded image -> token
attention the k dimension 48 -> 64 of P x V' and the structural zeros of the block-diagonal token-side weights).  The two must agree within 2 %: nothing
that does not run is credited (the history projection of a last window), nothing that runs for padding either."""
import os
import re

import bench
from l4p_amd.weights import ModelCfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILE = os.path.join(ROOT, "profiles", "r04_c3_per_shape_event_profile.txt")
LINE = re.compile(r"^(gemm|gemm_small|conv3d)\s+(M\d+ N\d+ K\d+ epi\d act\d .*?)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s*$")


def executed_flops():
    """Σ 2*M*N*K over the profile's dense launches, padding removed by bench.executed_flops_of_tag (the rule bench.py itself applies
    to the "gemm_small" class at run time); the small / streaming products count towards the model's "gemm" figure."""
    tot = {"gemm": 0.0, "conv3d": 0.0}
    for ln in open(PROFILE):
        m = LINE.match(ln.rstrip())
        if not m:
            continue
        cls, tag, n = m.group(1), m.group(2), float(m.group(3))
        tot["conv3d" if cls == "conv3d" else "gemm"] += bench.executed_flops_of_tag(tag) * n
    return tot


def test_algorithmic_flops_match_the_executed_shapes():
    B, nq = 4, 64
    fl = bench.algorithmic_flops(ModelCfg.full(), bench.ALL_TASKS, nq)
    ex = executed_flops()
    for cls in ("gemm", "conv3d"):
        want, got = fl[cls] * B, ex[cls]
        assert abs(got - want) <= 0.02 * want, (cls, got / 1e12, want / 1e12)


def test_history_projection_is_credited_only_where_it_is_needed():
    cfg = ModelCfg.full()
    S, D = cfg.tokens, cfg.dim
    one = bench.algorithmic_flops(cfg, ["track_2d"], 64, n_windows=1)["gemm"]
    three = bench.algorithmic_flops(cfg, ["track_2d"], 64, n_windows=3)["gemm"]  # per-window average
    # a 3-window clip vs three single-window clips: two windows project the half of the tokens their successor keeps
    # (2 * 0.5 * hist); the FIRST window shares the image-side projections of layer 0 across tracks in full, the two later
    # windows share their track-independent temporal half (2 * 63 * shared lost, 2 * 0.5 * 63 * shared of it recovered)
    hist, shared = 2.0 * S * D * D, 2 * 2.0 * S * D * (D // 2)  # (shared: t2i.k, t2i.v of layer 0; i2t is folded)
    assert abs((3 * three - 3 * one) - (64 * hist + 2 * 63 * shared - 63 * shared)) <= 1e-6 * one
    # a single window credits less than the reference graph's 73.81 GF per query (no history projection, folded i2t)
    assert bench.algorithmic_flops(cfg, ["track_2d"], 64)["gemm"] - bench.algorithmic_flops(cfg, ["track_2d"], 0)["gemm"] < 73.81e9 * 64 - 64 * hist + 1


def test_executed_flops_of_tag_padding_rules():
    """The per-launch FLOP count bench.py takes from a profiler tag: padding the kernels add is not counted."""
    f = bench.executed_flops_of_tag
    assert f("M8192 N6144 K1408 epi0 act1 8p t256x256") == 2.0 * 8192 * 6144 * 1408
    assert f("M8192 N4608 K1408 epi1 act0 8p t256x192") == 2.0 * 8192 * (4608 * 88 // 96) * 1408      # QKV: head dim 88 of 96
    assert f("M1048576 N768 K352 epi3 act1 8p t256x256") == 2.0 * 1048576 * (768 * 176 // 192) * 352  # 176 of 192 channels per tap
    assert f("M8192 N1408 K1216 epi0 act0 8p t256x256") == 2.0 * 8192 * 1408 * 1176                   # patch vector 1176 of 1216
    assert f("M131072 N1408 K64 epi0 act0 delta t16x128 wgrp") == 2.0 * 131072 * 1408 * 48            # 48 of 64 k slots
    assert f("M384 N11264 K704 epi0 act0 sk1 t128x64") == 2.0 * 384 * 1408 * 704                      # block-diagonal token-side weights
    assert f("M384 N8 K704 epi0 act0 sk1 t128x64 deep") == 2.0 * 384 * 1 * 704
    assert f("M3072 N1408 K2048 epi0 act0 ctx t48x128") == 2.0 * 3072 * 1408 * 2048
    assert f("group of 3: M384 N1408 K1408 ... t128x64 deep") == 0.0                                  # (grouped launches carry no shape)


def test_kernel_tree_hash_is_stable_and_tied_to_the_committed_traffic_file():
    """roofline.traffic is attached only from a PMC file taken on the kernel sources being run: the committed r04 file carries the
    hash of the committed tree."""
    import json

    from l4p_amd import _lib

    h = _lib.kernel_tree_hash()
    assert len(h) == 16 and h == _lib.kernel_tree_hash()
    import glob

    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_hbm_traffic.json")))[-1]  # (the file bench.py would attach)
    tj = json.load(open(latest))
    assert tj.get("kernel_tree") == h, f"{latest} was taken on other kernel sources: re-run tools/make_profiles.sh and commit its summaries"
    assert {"gemm", "gemm_small", "conv3d", "attention", "layernorm"} <= set(tj["per_class_bytes_per_launch"])
