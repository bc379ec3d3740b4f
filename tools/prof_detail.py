"""Per-shape kernel time table for one bench workload (event profiler, l4p_prof_detail).
usage: python tools/prof_detail.py [c2|c3|c3b8|c5|prep|demo] [steps] [c5: queries, default 64]   (demo: 64 frames, 625 queries, depth+flow+mask+tracks;
c3b8: all heads at batch 8, the per-GPU batch of configs[3] / of every rank of `bench.py --gpus N`)"""
import contextlib
import ctypes as C
import os
import sys

os.environ["L4P_TRACK_STREAMS"] = os.environ["L4P_HEAD_STREAMS"] = "0"  # serialise: overlapping streams inflate per-kernel event durations

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from l4p_amd import _lib


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib = _lib.load()
    if wl == "prep":  # the clip-preparation workload of bench.py --workload prep
        from l4p_amd.data import prepare_clip
        from tests.golden_utils import synthetic_video

        frames = torch.from_numpy(synthetic_video(100, 50, 480, 854)).to(dev)

        def run():
            return prepare_clip(frames, (64, 224, 224), (224, 224), spacing=0.04)
    elif wl == "demo":  # the generic-video case of demo/demo.py: one 64-frame clip, 625 grid queries in chunks of 128
        import time

        from l4p_amd.data import prepare_clip
        from tests.golden_utils import synthetic_video

        tasks = ["depth", "flow_2d_backward", "dyn_mask", "track_2d"]
        model, _, _ = bench.build_workload(list(bench.ALL_TASKS), 1, 64, dev)
        model.l4p_model.task_heads["track_2d"].max_queries = 128
        clip = prepare_clip(torch.from_numpy(synthetic_video(1, 50, 480, 854)).to(dev), (64, 224, 224), (224, 224), spacing=0.04)
        data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in clip.items()}

        def run():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                out = model.forward(data, tasks)
            torch.cuda.synchronize()
            print(f"demo forward: {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr)
            return out
    elif wl == "c5":  # one 256-frame video, all heads, one rank (bench.py --workload c5)
        from l4p_amd.parallel import forward_windows_sharded

        tasks = list(bench.ALL_TASKS)
        model, data, _ = bench.build_workload(tasks, 1, int(sys.argv[3]) if len(sys.argv) > 3 else 64, dev, frames=256)
        model.l4p_model.always_use_windowed_version = True

        def run():
            with torch.no_grad():
                return forward_windows_sharded(model.l4p_model, data, tasks, 0, 1, group=4)
    else:
        tasks = ["depth"] if wl == "c2" else list(bench.ALL_TASKS)
        B = 1 if wl == "c2" else (8 if wl == "c3b8" else 4)
        model, data, _ = bench.build_workload(tasks, B, 64, dev)

        def run():
            with torch.no_grad():
                with contextlib.redirect_stdout(sys.stderr):  # (the model mirrors the reference's stdout messages)
                    return model.forward(data, tasks)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    lib.l4p_prof_reset()
    lib.l4p_prof_enable(1)
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    lib.l4p_prof_enable(0)
    n = lib.l4p_prof_detail(None, 0)
    buf = C.create_string_buffer(int(n))
    lib.l4p_prof_detail(buf, n)
    tot = 0.0
    print(f"{'class':12s} {'tag':48s} {'n/step':>7s} {'ms/step':>9s} {'us/launch':>10s}")
    for line in buf.value.decode().splitlines():
        cls, tag, cnt, ms = line.split("\t")
        cnt, ms = int(cnt) / steps, float(ms) / steps
        tot += ms
        print(f"{cls:12s} {tag:48s} {cnt:7.1f} {ms:9.3f} {ms / cnt * 1e3:10.1f}")
    print(f"total {tot:.3f} ms/step")


if __name__ == "__main__":
    main()
