"""Victim: pointmap kernel (main stream).  Aggressor: the tracker recursion on a side stream, Python path (kernel by kernel), with
ONE kernel family at a time replaced by a no-op (its outputs are garbage: irrelevant here).  Removing which family removes the
victim's mismatches?  (round-4 race diagnosis)"""
import os
import sys

os.environ["L4P_TRACK_PYTHON"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "race_c5c.py")).read().split("def once():")[0])
from l4p_amd import parallel
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build


def victim():
    a = torch.empty(n, 3, device=dev)
    _lib.check(lib.l4p_point_map_samples(_stream(), _p(depth), _p(K), _p(P), _p(a), F, H, W, ratio, seed), "p")
    return a


cfg = ModelCfg.mini()
model = build(cfg, seeded_state_dict(cfg), "bf16")
net = model.l4p_model
batch = make_batch(256, 2)
data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
FAMS = ["l4p_track_tokens", "l4p_cast", "l4p_track_keys_init", "l4p_gemm", "l4p_small_attn", "l4p_layernorm_ex", "l4p_layernorm_res",
        "l4p_broadcast_block", "l4p_fill_rows", "l4p_layernorm_t", "l4p_mask_gather", "l4p_track_readout", "l4p_track_prepare",
        "l4p_track_commit"]
orig = {f: getattr(lib, f) for f in FAMS}
with torch.no_grad():
    groups = parallel.encode_local_windows(net, data, ["track_2d"], 0, 1, 8)
    lasts = parallel.local_last_features(groups, 1)
    wins = [parallel.DecodedWindow(net.cfg.depth, {}, lasts[w]["last"]) for w in range(31)]
    trk = net.task_heads["track_2d"]
    strides = net.time_strides(256)
    ref = victim().clone()
    torch.cuda.synchronize()
    for skip in [None] + FAMS + ["ALL"]:
        for f in FAMS:
            setattr(lib, f, (lambda *a: 0) if (f == skip or skip == "ALL") else orig[f])
        bad = runs = 0
        for rep in range(4):
            trk.defer_join = True
            keep = trk.forward_windowed(enc_features_bpc_2dlist=wins, time_strides=strides, **data)
            res = [victim() for _ in range(80)]
            trk.join_streams()
            trk.defer_join = False
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(r, ref)) for r in res)
            runs += len(res)
            del keep, res
        print(f"tracker without {str(skip):24s}: victim mismatches {bad} / {runs}")
