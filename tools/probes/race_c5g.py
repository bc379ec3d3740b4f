"""Does the tracker recursion (side stream) write into memory of the MAIN stream's allocator pool?  The main pool's cached free
blocks are taken as canaries (byte pattern), the tracker runs with its outputs kept alive, the canaries are checked."""
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
model = build(cfg, seeded_state_dict(cfg), "bf16")
net = model.l4p_model
batch = make_batch(256, 2)
data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
strides = net.time_strides(256)
with torch.no_grad():
    local = parallel.decode_local_windows(net, data, TASKS, 0, 1)
    windows = [parallel.DecodedWindow(net.cfg.depth, {k[4:]: v for k, v in local[w].items() if k.startswith("dec.")}, None) for w in range(31)]
    wins_t = [parallel.DecodedWindow(net.cfg.depth, {}, local[w]["last"]) for w in range(31)]
    trk = net.task_heads["track_2d"]
    o = net.stitch_windows(windows, data, DENSE, strides)  # populates the main pool's free lists
    del o
    torch.cuda.synchronize()
    reserved = torch.cuda.memory_reserved()
    canaries = []
    for size in [256 << 20, 64 << 20, 16 << 20, 4 << 20, 1 << 20, 256 << 10, 64 << 10, 16 << 10, 4 << 10, 512]:
        while True:
            t = torch.empty(size, dtype=torch.uint8, device="cuda")
            if torch.cuda.memory_reserved() > reserved:  # came from a fresh segment: the cached blocks of this size are used up
                del t
                torch.cuda.empty_cache() if False else None
                reserved = torch.cuda.memory_reserved()
                break
            canaries.append(t)
            if len(canaries) > 20000:
                break
    for t in canaries:
        t.fill_(0xAB)
    torch.cuda.synchronize()
    print("canaries:", len(canaries), "bytes", sum(t.numel() for t in canaries))
    trk.defer_join = True
    keep = trk.forward_windowed(enc_features_bpc_2dlist=wins_t, time_strides=strides, **data)
    trk.join_streams()
    trk.defer_join = False
    torch.cuda.synchronize()
    hit = 0
    for t in canaries:
        bad = (t != 0xAB)
        if bool(bad.any()):
            idx = bad.nonzero().flatten()
            hit += 1
            if hit <= 8:
                print("canary of", t.numel(), "bytes at", hex(t.data_ptr()), ":", int(bad.sum()), "bytes changed, offsets", int(idx[0]), "..", int(idx[-1]),
                      "as floats:", t[int(idx[0]) // 4 * 4:int(idx[0]) // 4 * 4 + 16].view(torch.float32).tolist())
    print("canaries written by the tracker:", hit)
