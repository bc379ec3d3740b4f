"""Dense stitch of 31 pre-decoded windows (main stream) beside (a) nothing, (b) torch matmuls on a side stream, (c) the tracker
recursion on a side stream with its outputs kept alive.  Which combinations are reproducible?  (round-4 race diagnosis)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from l4p_amd import parallel
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build

TASKS = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]
DENSE = [t for t in TASKS if t != "track_2d"]
cfg = ModelCfg.mini()
model = build(cfg, seeded_state_dict(cfg), sys.argv[1] if len(sys.argv) > 1 else "bf16")
net = model.l4p_model
batch = make_batch(256, 2)
data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
strides = net.time_strides(256)
side = torch.cuda.Stream()
x = torch.randn(2048, 2048, device="cuda")
with torch.no_grad():
    local = parallel.decode_local_windows(net, data, TASKS, 0, 1)
    windows = [parallel.DecodedWindow(net.cfg.depth, {k[4:]: v for k, v in local[w].items() if k.startswith("dec.")}, None) for w in range(31)]
    wins_t = [parallel.DecodedWindow(net.cfg.depth, {}, local[w]["last"]) for w in range(31)]
    trk = net.task_heads["track_2d"]
    torch.cuda.synchronize()

    def dense():
        o = net.stitch_windows(windows, data, DENSE, strides)
        return {k: v for k, v in o.items() if torch.is_tensor(v)}

    base = dense()
    torch.cuda.synchronize()
    for mode in ("alone", "beside torch matmuls", "beside the tracker", "alone"):
        nbad = 0
        for rep in range(5):
            keep = None
            if mode == "beside torch matmuls":
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(400):
                        keep = torch.nn.functional.gelu(x @ x)
            elif mode == "beside the tracker":
                trk.defer_join = True
                keep = trk.forward_windowed(enc_features_bpc_2dlist=wins_t, time_strides=strides, **data)
            cur = dense()
            if mode == "beside the tracker":
                trk.join_streams()
                trk.defer_join = False
            torch.cuda.synchronize()
            bad = [k for k in base if not torch.equal(base[k], cur[k])]
            nbad += bool(bad)
            if bad:
                print("   ", mode, "rep", rep, "differing:", bad)
            del keep
        print(mode, ":", nbad, "of 5 stitches differ")
