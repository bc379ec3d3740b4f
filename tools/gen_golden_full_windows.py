"""Full-size (VideoMAE-v2-giant geometry) TWO-window golden from the REAL reference (runs only where /root/reference exists):
24 frames = 2 overlapping windows, depth (inverse-depth LstSq seam), backward flow (first frame of the later window skipped),
motion mask (overwrite) and 4 tracks (memory tokens + re-seeding across the seam) - the windowed path of configs[4] at the real
geometry.  Writes tests/golden/full_T24_windows.npz (sampled values only).

  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_full_windows.py        (~5 minutes on 8 cores)"""
from __future__ import annotations

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
from tests.golden_utils import make_batch, sample_indices
from tools.gen_golden import build_reference, install_stubs

TASKS = ["depth", "flow_2d_backward", "dyn_mask", "track_2d"]


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
    batch = make_batch(24, 4)
    t0 = time.time()
    with torch.no_grad():
        out = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
    print(f"reference forward: {time.time() - t0:.1f}s", flush=True)
    npz = {}
    for k, v in out.items():
        if not torch.is_tensor(v):
            continue
        v = v.detach().float()
        npz[k] = v.reshape(-1)[sample_indices(v.numel())].numpy() if v.numel() > 4096 else v.numpy()
        print(k, tuple(v.shape), float(v.abs().max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "full_T24_windows.npz"), **npz)


if __name__ == "__main__":
    main()
