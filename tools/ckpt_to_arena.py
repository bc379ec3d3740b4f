"""Offline checkpoint ingest: reference Lightning checkpoint -> packed weight arena.

  python tools/ckpt_to_arena.py weights/l4p_depth_flow_2d3dtrack_camray_dynseg_v1.ckpt weights/l4p.f16.l4parena
         [--precision 16-mixed|bf16|32-true] [--tasks depth,flow_2d_backward,...] [--mini]

--precision defaults to "16-mixed" - the default of prepare_model / build_model and what demo/demo.py asks for (the reference's Fabric
"16-mixed" = IEEE half) - so an arena packed with the tool's defaults is the one prepare_model's defaults accept.  The strings are the
ones l4p_amd.models.l4p_videomae accepts (one mapping, `_engine_dtype`); anything else is an error, not float32.

Reads {"state_dict": {916 keys prefixed "l4p_model."}} (l4p/models/utils.py:52-53), checks it strictly against the
schema (l4p_amd.weights.state_dict_schema == the reference's key set and shapes), repacks every tensor into the kernel
layouts (l4p_amd/packing.py) and writes ONE file: magic, JSON header (layout, geometry, dtype) and the raw arena — the
bytes `prepare_model(..., ckpt_path="x.l4parena")` uploads as they are (and what rank 0 broadcasts over RCCL).  Needs
no GPU."""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from l4p_amd._lib import L4P_BF16, L4P_F16, L4P_F32
from l4p_amd.models.l4p_videomae import _engine_dtype
from l4p_amd.packing import pack_state_dict
from l4p_amd.weights import ModelCfg, state_dict_schema


def convert(ckpt_path: str, out_path: str, precision: str = "16-mixed", tasks=None, cfg: ModelCfg = None) -> dict:
    cfg = cfg or ModelCfg.full()
    sd = torch.load(ckpt_path, weights_only=True, map_location="cpu")
    sd = sd["state_dict"] if "state_dict" in sd else sd
    sd = {(k[len("l4p_model."):] if k.startswith("l4p_model.") else k): v for k, v in sd.items()}
    exp = state_dict_schema(cfg, tasks=tasks)
    missing = [k for k in exp if k not in sd]
    bad = [k for k in exp if k in sd and tuple(sd[k].shape) != tuple(exp[k])]
    if missing or bad:
        raise SystemExit(f"checkpoint does not match the model schema: missing={missing[:5]} ({len(missing)}), "
                         f"shape mismatch={bad[:5]} ({len(bad)})")
    td = {L4P_BF16: torch.bfloat16, L4P_F16: torch.float16, L4P_F32: torch.float32}[_engine_dtype(precision)]  # (raises on unknown strings)
    pw = pack_state_dict(sd, cfg, td, torch.device("cpu"), tasks=tasks)
    extra = {"dtype": {torch.bfloat16: "bfloat16", torch.float16: "float16", torch.float32: "float32"}[td], "geometry": cfg.describe(),
             "tasks": list(tasks) if tasks else None, "source": os.path.basename(ckpt_path)}
    pw.save(out_path, extra)
    return {"tensors": len(pw.layout), "bytes": int(pw.arena.numel()), **extra}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ckpt")
    ap.add_argument("out")
    ap.add_argument("--precision", default="16-mixed", help="16-mixed (default, = prepare_model's) | bf16 | 32-true")
    ap.add_argument("--tasks", default=None, help="comma-separated subset of heads to pack (default: all in the checkpoint)")
    ap.add_argument("--mini", action="store_true", help="the tests' 704-wide / 4-deep geometry")
    a = ap.parse_args()
    info = convert(a.ckpt, a.out, a.precision, a.tasks.split(",") if a.tasks else None, ModelCfg.mini() if a.mini else None)
    print(info)


if __name__ == "__main__":
    main()
