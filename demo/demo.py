"""The generic-video part of the reference's demo (demo/demo.py:80-112) on the MI355X engine: VideoDataset ->
DataLoader(batch_size=1) -> prepare_model(...).forward(batch, tasks).  Visualisation (l4p/utils/vis.py, viser) is out of
scope; the outputs are reported (and optionally saved as .npz) instead.

  python demo/demo.py --videos a.mp4 b.mp4 --ckpt weights/l4p_depth_flow_2d3dtrack_camray_dynseg_v1.ckpt   # needs mediapy
  python demo/demo.py --synthetic                      # no checkpoint / video files here: seeded weights + a seeded video

With --synthetic the weights are the name-seeded random tensors of the test-suite (same 916-key state dict a checkpoint
holds) and the "video" is l4p_amd.data.synthetic.synthetic_video: the point is the plumbing and the timing, not the pictures.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from l4p_amd.data import VideoDataset
from l4p_amd.models.utils import build_model, prepare_model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", nargs="*", default=[])
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "model.yaml"))
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--frames", type=int, default=64, help="crop_size[0] (the demo uses 64, or 16 to limit memory)")
    ap.add_argument("--max-queries", type=int, default=128)
    ap.add_argument("--spacing", type=float, default=0.04, help="track_2d_querry_sampling_spacing (625 queries at 0.04)")
    ap.add_argument("--save", default=None, help="directory for <seq_name>.npz")
    ap.add_argument("--precision", default="16-mixed",
                    help="engine: 16-mixed (the reference demo's own, IEEE half; default) | bf16 (what bench.py measures) | 32-true")
    args = ap.parse_args()

    precision, accelerator = args.precision, "gpu"  # demo.py:22-23 hard-codes "16-mixed"
    tasks = ["depth", "flow_2d_backward", "dyn_mask", "track_2d"]  # demo.py:82,99
    frames = None
    if args.synthetic:
        from l4p_amd.weights import ModelCfg, seeded_state_dict
        from l4p_amd.data.synthetic import synthetic_video

        model = build_model(args.config, max_queries=args.max_queries, precision=precision)
        model.load_state_dict({"l4p_model." + k: v for k, v in seeded_state_dict(ModelCfg.full()).items()})
        model = model.eval()
        args.videos = ["synthetic/480p.mp4"]
        frames = {args.videos[0]: synthetic_video(1, 50, 480, 854)}
    else:
        assert args.ckpt and args.videos, "--ckpt and --videos (or --synthetic)"
        model = prepare_model(model_config_path=args.config, ckpt_path=args.ckpt, max_queries=args.max_queries,
                              precision=precision, accelerator=accelerator)

    model.l4p_model.window_batch = 8  # windows of a long clip go through encoder + dense decoders eight at a time
    dataset = VideoDataset(video_paths=args.videos, crop_size=(args.frames, 224, 224), estimation_directions=[1],
                           track_2d_querry_sampling_spacing=args.spacing, frames=frames)
    loader = torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False)
    for batch in loader:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = model.forward(batch, tasks)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        T = batch["rgb_b3thw"].shape[2]
        print(f"{batch['seq_name'][0]}: {T} frames, {batch['track_2d_pointquerries_bn3'].shape[1]} queries, "
              f"{dt * 1e3:.1f} ms ({T / dt:.1f} frames/s)")
        for k, v in out.items():
            if torch.is_tensor(v):
                print(f"  {k:32s} {tuple(v.shape)} {v.dtype}")
        if args.save:
            os.makedirs(args.save, exist_ok=True)
            np.savez_compressed(os.path.join(args.save, os.path.splitext(batch["seq_name"][0])[0] + ".npz"),
                                **{k: v.float().cpu().numpy() for k, v in out.items() if torch.is_tensor(v)})


if __name__ == "__main__":
    main()
