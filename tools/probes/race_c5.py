"""Which outputs of the long-video forward change when the tracker runs on its own stream beside the dense work (L4P_TRACK_DEFER=1)
instead of in front of it (=0)?  Mini geometry, 256 frames, bf16.  (round-4 diagnosis of a stream race)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from l4p_amd import parallel
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build

TASKS = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
cfg = ModelCfg.mini()
model = build(cfg, seeded_state_dict(cfg), prec)
net = model.l4p_model
batch = make_batch(256, 11)


def fwd(defer):
    os.environ["L4P_TRACK_DEFER"] = defer
    with torch.no_grad():
        o = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}


a = fwd("0")
for trial in range(3):
    b = fwd("1")
    bad = {k: float((a[k] - b[k]).abs().max()) for k in a if not torch.equal(a[k], b[k])}
    print("forward defer=1 vs defer=0, trial", trial, "differing keys:", bad)
    for k in bad:
        d = (a[k] != b[k])
        tdim = 2 if a[k].dim() >= 3 else -1
        idx = d.nonzero()
        print("   ", k, "first differing index", idx[0].tolist(), "count", int(d.sum()), "frames", sorted(set(idx[:, tdim].tolist()))[:6] if a[k].dim() >= 3 else "")
c = fwd("0")
print("defer=0 repeat equal:", all(torch.equal(a[k], c[k]) for k in a))

# ---- the emulated-rank path (tests/test_sharded_windows_gpu.py) ----
data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
world = 8
with torch.no_grad():
    merged = {}
    for r in range(world):
        merged.update(parallel.decode_local_windows(net, data, TASKS, r, world))
    gathered = [merged[w] for w in range(31)]
    torch.cuda.synchronize()

    def stitch(defer, r, sync=False):
        os.environ["L4P_TRACK_DEFER"] = defer
        o = parallel.stitch_gathered_windows(net, data, TASKS, gathered, r, world)
        if sync:
            torch.cuda.synchronize()
        return {k: v for k, v in o.items() if torch.is_tensor(v)}

    base = [stitch("0", r, True) for r in range(world)]
    torch.cuda.synchronize()
    for mode in ("sync after every rank", "no sync between ranks"):
        outs = [stitch("1", r, mode.startswith("sync")) for r in range(world)]
        torch.cuda.synchronize()
        for r in range(world):
            bad = {k: float((base[r][k] - outs[r][k]).abs().max()) for k in base[r] if not torch.equal(base[r][k], outs[r][k])}
            print(mode, "rank", r, "differing keys:", bad)
    for r in range(world):
        bad = [k for k in a if k in base[r] and not k.startswith("track") and not torch.equal(a[k], base[r][k])]
        print("defer=0 emulated rank", r, "vs forward: differing dense keys", bad)
