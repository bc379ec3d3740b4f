"""Evaluation only ("fp8 keys"): what would storing the tracker's key-side projections in fp8 cost in accuracy?

The image-side operands of the SAM-style decoder's cross attentions (t2i.k, t2i.v, i2t.q of both layers and final.k / final.v:
[N * 2048, 704] per query chunk, the tensors the small-attention kernels stream from HBM) are rounded to float8 e4m3 and back
right after the projection that produces them — the values an fp8 store would hold — in the Python composition of the tracker
window (L4P_TRACK_PYTHON=1; the native window is the same kernels).  Everything else runs as shipped (bf16 engine).  Reported:
trajectory / visibility / depth differences against the unmodified bf16 engine and against the f32 engine, on the full-size model
with name-seeded weights, 24 frames = 2 windows, 16 queries.  No fp8 kernel exists; this sizes the decision."""
import os
import sys

os.environ["L4P_TRACK_PYTHON"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from l4p_amd.models.task_heads import sparse_heads as sh
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build

KEYS = ("t2i.k", "t2i.v", "i2t.q", "final.k", "final.v")


def run(model, batch):
    with torch.no_grad():
        out = model.forward({k: v.clone() for k, v in batch.items()}, ["track_2d"])
    torch.cuda.synchronize()
    return {k: v.float().cpu() for k, v in out.items() if k.startswith("track_2d") and torch.is_tensor(v)}


def main():
    full = "--mini" not in sys.argv
    cfg = ModelCfg.full() if full else ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    batch = make_batch(24, 16)
    ref32 = run(build(cfg, sd, "32-true"), batch)
    model = build(cfg, sd, "bf16")
    base = run(model, batch)
    orig = sh.VideoMAETrack2DSamHead._proj
    stats = {"n": 0}

    def proj_fp8(self, x, key, n, **kw):
        y = orig(self, x, key, n, **kw)
        if key.endswith(KEYS) and y.shape[0] >= 2048:  # image-side operands only (the prompt-token side has 6 rows per query)
            stats["n"] += 1
            y.copy_(y.to(torch.float8_e4m3fn).to(y.dtype))
        return y

    sh.VideoMAETrack2DSamHead._proj = proj_fp8
    try:
        q8 = run(model, batch)
    finally:
        sh.VideoMAETrack2DSamHead._proj = orig
    print(f"model: {'full' if full else 'mini'}; projections rounded to e4m3: {stats['n']}")
    for k in sorted(base):
        a, b, r = base[k], q8[k], ref32[k]
        rl = lambda x, y: float((x - y).norm() / y.norm().clamp_min(1e-12))  # noqa: E731
        extra = ""
        if "traj" in k:
            extra = f"  max |dx| bf16->fp8 {float((a - b).abs().max()):.3f} px, bf16->f32 {float((a - r).abs().max()):.3f} px"
        if "vis" in k:
            extra = f"  visibility sign flips bf16->fp8 {int(((a > 0) != (b > 0)).sum())} / {a.numel()}, bf16->f32 {int(((a > 0) != (r > 0)).sum())}"
        print(f"{k}: rel-L2 fp8 vs bf16 {rl(b, a):.2e} | bf16 vs f32 {rl(a, r):.2e} | fp8 vs f32 {rl(b, r):.2e}{extra}")


if __name__ == "__main__":
    main()
