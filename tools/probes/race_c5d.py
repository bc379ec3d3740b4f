"""Which INPUT of the seam RANSAC differs when the tracker runs beside the dense stitch?  (round-4 race diagnosis)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from l4p_amd import _lib, parallel
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
lib = _lib.load()
log = []
orig_r, orig_q, orig_p = lib.l4p_similarity_ransac, lib.l4p_quantile, lib.l4p_point_map_samples


def view(ptr, n, dtype=torch.float32):
    # a tensor over raw device memory (read right after the launch, on the launch stream)
    buf = (C.c_char * 0).from_address(0)
    t = torch.empty(n, dtype=dtype, device="cuda")
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(n * t.element_size()), 3,
                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return t


def ransac(stream, src, dst, n, q98, thr, trials, ms, seed, ws, out):
    rc = orig_r(stream, src, dst, n, q98, thr, trials, ms, seed, ws, out)
    g = lambda p: p.value if hasattr(p, "value") else int(p)
    log.append({"src": view(g(src), n * 3), "dst": view(g(dst), n * 3), "q98": view(g(q98), 1), "scores": view(g(ws), 2 * trials),
                "out": view(g(out), 18)})
    return rc


lib.l4p_similarity_ransac = ransac
with torch.no_grad():
    merged = {}
    for r in range(world):
        merged.update(parallel.decode_local_windows(net, data, TASKS, r, world))
    gathered = [merged[w] for w in range(31)]
    torch.cuda.synchronize()
    for defer in ("0", "1", "1", "1"):
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
                nd = {k: int((a[k] != b[k]).sum()) for k in bad}
                print("defer=1: first differing seam", s + 1, "differing element counts", nd, "q98", a["q98"].item(), b["q98"].item())
                for k in ("src", "dst"):
                    if k in bad:
                        idx = (a[k] != b[k]).nonzero().flatten()
                        print("   ", k, "first/last differing flat index", int(idx[0]), int(idx[-1]), "of", a[k].numel(), "values", a[k][idx[:3]].tolist(), b[k][idx[:3]].tolist())
                break
        else:
            print("defer=1: no difference in", len(base), "seams")
