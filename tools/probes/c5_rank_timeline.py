"""One of eight ranks of configs[4] as bench.c5_phase_times emulates it, in three separated segments for a rocprofv3 --kernel-trace:
  A  encoder of rank 0's windows, then its decoders BESIDE the tracker on an eighth of the queries (the emulated rank)
  B  the tracker on an eighth of the queries alone
  C  the decoders of rank 0's windows alone
Segments are separated by 0.5 s of idle GPU; tools/probes/c5_rank_timeline_report.py cuts the trace there and reports, per segment,
the span, the busy time per hardware queue and how much of it overlaps.
usage: rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/probes/c5_rank_timeline.py"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from l4p_amd import parallel as par


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tasks = list(bench.ALL_TASKS)
    model, data, _ = bench.build_workload(tasks, 1, 64, dev, frames=256)
    net = model.l4p_model
    net.always_use_windowed_version = True
    strides = net.time_strides(data["rgb_b3thw"].shape[2])
    nwin = len(strides)
    B = 1
    group = 4

    def run_tracker(lasts, r, w):
        d, n = par.shard_track_inputs(data, r, w)
        wins = [par.DecodedWindow(net.cfg.depth, {}, g["last"]) for g in lasts]
        return net.task_heads["track_2d"].forward_windowed(enc_features_bpc_2dlist=wins, time_strides=strides, **d)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) * 1e3

    with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
        groups = par.encode_local_windows(net, data, tasks, 0, 1, group)
        lasts = par.all_gather_windows(par.local_last_features(groups, B), nwin, 0, 1)
        del groups
        torch.cuda.synchronize()

        order = os.environ.get("C5_TL_ORDER", "dec_first")
        host = {}

        def seg_a():
            h0 = time.perf_counter()
            g8 = par.encode_local_windows(net, data, tasks, 0, 8, group)
            h1 = time.perf_counter()
            tr = net.task_heads["track_2d"]
            tr.defer_join = tr.own_stream = True
            tr.start_event = torch.cuda.Event()
            tr.start_event.record(torch.cuda.current_stream())
            try:
                if order == "dec_first":
                    # C5_TL_DEC_CUS="first,count": the decoders on a stream confined to those CUs (l4p_stream_create_cu_mask)
                    # (default: parallel.decoder_stream's choice for an 8-query shard)
                    spec = os.environ.get("C5_TL_DEC_CUS")
                    ds = par.decoder_stream(dev, 8) if spec is None else par.cu_masked_stream(dev, spec)
                    par.decode_encoded_windows_on(ds, net, data, tasks, g8)
                    h2 = time.perf_counter()
                    o = run_tracker(lasts, 0, 8)
                    h3 = time.perf_counter()
                elif order == "thread":  # the recursion issued by its own host thread while this one issues the decoders
                    import threading

                    box = {}

                    def work():
                        torch.cuda.set_device(dev)
                        with torch.no_grad():
                            box["o"] = run_tracker(lasts, 0, 8)
                        box["t"] = time.perf_counter()

                    th = threading.Thread(target=work)
                    th.start()
                    par.decode_encoded_windows(net, data, tasks, g8)
                    h2 = time.perf_counter()
                    th.join()
                    o = box["o"]
                    h3 = box["t"]
                else:
                    o = run_tracker(lasts, 0, 8)
                    h2 = time.perf_counter()
                    par.decode_encoded_windows(net, data, tasks, g8)
                    h3 = time.perf_counter()
                host.update(enc=(h1 - h0) * 1e3, second=(h2 - h1) * 1e3, third=(h3 - h2) * 1e3)
                tr.join_streams()
            finally:
                tr.join_streams()
                tr.defer_join = tr.own_stream = False
                tr.start_event = None
            return o

        def seg_b():
            return run_tracker(lasts, 0, 8)

        g8c = par.encode_local_windows(net, data, tasks, 0, 8, group)

        def seg_c():
            return par.decode_encoded_windows(net, data, tasks, g8c)

        for f in (seg_a, seg_b, seg_c):
            f()
            f()
        torch.cuda.synchronize()
        for name, f in (("A emulated rank (encoder, decoders beside the tracker)", seg_a), ("B tracker alone", seg_b), ("C decoders alone", seg_c)):
            time.sleep(0.5)
            _, ms = timed(f)
            print(f"segment {name}: {ms:.2f} ms", flush=True)
            print(f"segment {name}: {ms:.2f} ms", file=sys.stderr, flush=True)
            if f is seg_a:
                print(f"  host enqueue times ({order}): encoder {host['enc']:.2f} ms, then {host['second']:.2f} ms, then {host['third']:.2f} ms", flush=True)
        time.sleep(0.5)


if __name__ == "__main__":
    main()
