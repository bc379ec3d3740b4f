"""Which quantity of the joint seam alignment first differs when the tracker runs beside the dense stitch?  (round-4 race diagnosis)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from l4p_amd import parallel
from l4p_amd.utils import umeyama
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build

TASKS = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]
cfg = ModelCfg.mini()
model = build(cfg, seeded_state_dict(cfg), "bf16")
net = model.l4p_model
batch = make_batch(256, 11)
data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
world = 8
log = []
orig = umeyama.solve_window_similarity


def wrapped(cfg_obj, pred, target, img_info):
    rec = {k + ".p": pred[k].double().sum().clone() for k in pred}
    rec.update({k + ".t": target[k].double().sum().clone() for k in target})
    out = orig(cfg_obj, pred, target, img_info)
    rec["sim"] = out.clone()
    log.append(rec)
    return out


umeyama.solve_window_similarity = wrapped
with torch.no_grad():
    merged = {}
    for r in range(world):
        merged.update(parallel.decode_local_windows(net, data, TASKS, r, world))
    gathered = [merged[w] for w in range(31)]
    torch.cuda.synchronize()
    runs = {}
    for defer in ("0", "1", "1"):
        os.environ["L4P_TRACK_DEFER"] = defer
        log.clear()
        parallel.stitch_gathered_windows(net, data, TASKS, gathered, 0, world)
        torch.cuda.synchronize()
        cur = [{k: v.cpu() for k, v in rec.items()} for rec in log]
        if defer == "0":
            base = cur
            continue
        for s, (a, b) in enumerate(zip(base, cur)):
            bad = [k for k in a if not torch.equal(a[k], b[k])]
            if bad:
                print("defer=1: first differing seam", s + 1, "quantities", bad, {k: (a[k].flatten()[:4].tolist(), b[k].flatten()[:4].tolist()) for k in bad[:3]})
                break
        else:
            print("defer=1: no difference in", len(base), "seams")
