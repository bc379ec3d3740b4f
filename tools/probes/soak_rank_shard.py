"""Soak of a rank's query shard of configs[4]: the tracker on 8 of the 64 queries over all 31 windows (the one-wave projections, the
folded layer 0, the wide read-out, the grouped context product) beside the decoders of the rank's windows on their CU-masked stream,
N times over - tracks and decoded windows must be bit-identical to the first pass.  usage: python tools/probes/soak_rank_shard.py [iters]"""
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from l4p_amd import parallel as par


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tasks = list(bench.ALL_TASKS)
    model, data, _ = bench.build_workload(tasks, 1, 64, dev, frames=256)
    net = model.l4p_model
    net.always_use_windowed_version = True
    strides = net.time_strides(data["rgb_b3thw"].shape[2])
    nwin = len(strides)
    with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
        groups = par.encode_local_windows(net, data, tasks, 0, 1, 4)
        lasts = par.all_gather_windows(par.local_last_features(groups, 1), nwin, 0, 1)
        del groups
        d8, n8 = par.shard_track_inputs(data, 0, 8)
        wins = [par.DecodedWindow(net.cfg.depth, {}, g["last"]) for g in lasts]
        tr = net.task_heads["track_2d"]

        def once():
            g8 = par.encode_local_windows(net, data, tasks, 0, 8, 4)
            tr.defer_join = tr.own_stream = True
            tr.start_event = torch.cuda.Event()
            tr.start_event.record(torch.cuda.current_stream())
            try:
                local = par.decode_encoded_windows_on(par.decoder_stream(dev, n8), net, data, tasks, g8)
                o = tr.forward_windowed(enc_features_bpc_2dlist=wins, time_strides=strides, **d8)
                tr.join_streams()
            finally:
                tr.join_streams()
                tr.defer_join = tr.own_stream = False
                tr.start_event = None
            torch.cuda.synchronize()
            res = {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}
            for w, d in local.items():
                for k, v in d.items():
                    res[f"w{w}.{k}"] = v.clone()
            return res

        ref = once()
        bad = 0
        for i in range(iters):
            cur = once()
            diff = [k for k, v in ref.items() if not torch.equal(cur[k], v)]
            if diff:
                bad += 1
                print(f"iter {i}: differs in {diff[:6]}", flush=True)
    print(f"soak: {bad} of {iters} passes differ from the first ({len(ref)} tensors compared bitwise; {n8} queries, {nwin} windows)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
