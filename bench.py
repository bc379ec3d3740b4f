"""Throughput benchmark of the L4P hot path on MI355X (contract in the task statement).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one forward of the hot path (shared encoder + the workload's heads) over one batch of
synthetic 16x224x224 clips already resident in HBM.  One process per GPU, clips sharded with no
collective in the step (weak scaling: per-GPU batch fixed); the only collective is the one-off RCCL
broadcast of the packed weight arena before the timed region.  Rank 0 prints ONE JSON line.

--workload demo: the generic-video case of the reference's demo (demo/demo.py:84-100,38-40): one decoded 480x854 video per
GPU -> clip preparation -> 64 frames = 7 windows, tasks depth + flow + dyn_mask + track_2d, 625 grid queries (spacing 0.04)
tracked in chunks of max_queries = 128, all 7 windows batched through encoder and decoders (row f4 of SURVEY.md 8).

Workloads (BASELINE.json configs): c3 = all heads, bf16, batch 4 clips per GPU (DEFAULT: BASELINE.json's metric is
"frames/sec (all heads)", configs[2] is its single-GPU configuration and configs[3] the same work data-parallel over 8
GPUs); c2 = depth head only, bf16, batch 1 (configs[1]).
"""
from __future__ import annotations

import argparse
import contextlib
import ctypes as C
import json
import os
import re
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL across processes)

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from l4p_amd import _lib  # noqa: E402
from l4p_amd.models.utils import build_model  # noqa: E402
from l4p_amd.packing import pack_state_dict  # noqa: E402
from l4p_amd.parallel import broadcast_weights, collective_selftest, init_distributed  # noqa: E402
from l4p_amd.weights import ModelCfg, actpost_of, fusion_of, seeded_state_dict  # noqa: E402

PEAK_BF16_MFMA = 2.5e15  # dense bf16 / f16 MFMA peak, MI355X_MICROARCH.md
PEAK_F32_MFMA = 157.3e12  # f32-input MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32): the vector rate, 1/16 of the 16-bit peak
# --precision: the engine's arithmetic / storage type.  "bf16" is what BASELINE.json's configs name (the default, the driver's line);
# "16-mixed" = IEEE half, the mode the reference's own demo ships (Fabric "16-mixed" = fp16 autocast); "32-true" = the exact-f32
# parity engine, priced against the f32 MFMA peak.
PRECISION = {"bf16": ("bf16", torch.bfloat16, PEAK_BF16_MFMA), "16-mixed": ("f16", torch.float16, PEAK_BF16_MFMA),
             "32-true": ("f32", torch.float32, PEAK_F32_MFMA)}
ENGINE_PRECISION = "bf16"  # set from --precision in main()
ALL_TASKS = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]


TRACK_FLOPS_PER_QUERY_WINDOW = 73.81e9  # reference graph, SURVEY.md §2.1 / BASELINE.md §4 (incl. the history projection)


def algorithmic_flops(cfg: ModelCfg, tasks, n_queries: int, n_windows: int = 1):
    """FLOPs (2*MAC, padding excluded) of ONE 16-frame window per kernel class, from the model graph
    (SURVEY.md §2.1 / BASELINE.md §4), counting what the algorithm needs — work the reference executes and then discards
    is not credited (n_windows = windows of the clip the tracker recursion runs over; the per-window average is returned)."""
    S, D, Hd, H, dh = cfg.tokens, cfg.dim, cfg.mlp_hidden, cfg.heads, cfg.head_dim
    kraw = cfg.in_chans * cfg.patch[0] * cfg.patch[1] * cfg.patch[2]
    dense = [t for t in tasks if t in cfg.dense_tasks]
    depth_run = max(cfg.hooks) if dense and "track_2d" not in tasks else cfg.depth
    gemm = 2.0 * S * kraw * D + depth_run * 2.0 * S * (D * 3 * D + D * D + 2 * D * Hd)
    attn = depth_run * 4.0 * S * S * dh * H
    conv = 0.0
    nt, nh, nw = cfg.grid
    for t in dense:
        ap, fu = actpost_of(t), fusion_of(t)
        shapes = []
        for i in range(4):
            L = cfg.layer_dims[i]
            gemm += 2.0 * S * D * L
            g = [nt, nh, nw]
            if any(s > 0 for s in ap[i]):
                k = [2 ** s for s in ap[i]]
                gemm += 2.0 * S * L * L * k[0] * k[1] * k[2]
                g = [g[j] * k[j] for j in range(3)]
            elif any(s < 0 for s in ap[i]):
                g = [(g[j] - 1) // (2 ** (-ap[i][j])) + 1 for j in range(3)]
                conv += 2.0 * g[0] * g[1] * g[2] * 27 * L * L
            vox = g[0] * g[1] * g[2]
            conv += 2.0 * vox * 27 * L * cfg.feature_dim
            shapes.append(g)
        F_ = cfg.feature_dim
        cur = None
        for r, i in ((4, 3), (3, 2), (2, 1), (1, 0)):
            g = shapes[i]
            vox = g[0] * g[1] * g[2]
            n_rcu = 1 if r == 4 else 2
            conv += n_rcu * 2 * 2.0 * vox * 27 * F_ * F_
            gemm += 2.0 * vox * F_ * F_  # out_conv (applied before the up-sampling here)
            cur = [g[j] * fu[i][j] for j in range(3)]
        vox = cur[0] * cur[1] * cur[2]
        conv += 2.0 * vox * 27 * F_ * (F_ // 2)
        osz = (16, 16, 16) if t == "camray" else (cfg.frames, cfg.img, cfg.img)
        vo = osz[0] * osz[1] * osz[2]
        conv += 2.0 * vo * 27 * (F_ // 2) * cfg.last_dim
    # tracker: 73.81 GFLOP per query and window in the reference graph, all but ~1 % of it in projections / up-scaling
    # ConvTransposes that run through the GEMM kernel.  Three corrections, so that only work whose result is USED is credited:
    #  * processed_video_features_proj (sparse_heads.py:660-663, 2*S*D*D = 8.12 GF per query) feeds the NEXT window's
    #    memory tokens only, and only its second temporal half survives (sparse_heads.py:406-448): the last (or only)
    #    window needs none of it, every other window half of it;
    #  * in a first window every track starts from the same keys, so the first layer's image-side projections of the
    #    token -> image attention (t2i.k, t2i.v: 2*S*D*(D/2) each) are one computation per clip, not one per track; in later
    #    windows the second temporal half of the keys is still common to all tracks: half of those projections is one computation;
    #  * round 4: the image -> token attention is evaluated in its FOLDED form (packing.py fold_i2t: the projections of the S image
    #    tokens folded into the 6 prompt tokens of a track) - per layer the i2t.q and i2t.out projections (2*S*D*(D/2) each) and
    #    the attention between them (4*S*6*(D/2)) are replaced by scores = keys x K'^T and delta = P x V' (2*S*D*6*heads each) and
    #    the token-side products K' = k W_q, V' = W_out v (2*6*(D/2)*D each, block-diagonal over the heads).  The folded form
    #    is what an implementation needs to execute, so it is what is credited: 14.8 GF per query and window less, and 7.6 GF
    #    more for the two folded key projections (and again for the two folded value projections, below) of the token -> image attentions (layer 1 and the final one).
    if "track_2d" in tasks and n_queries > 0:
        hist_full = 2.0 * S * D * D
        HT = 6 * cfg.sam_heads
        i2t_projected = 2 * (2.0 * S * D * (D // 2)) + 4.0 * S * 6 * (D // 2)
        i2t_folded = 2 * (2.0 * S * D * HT) + 2 * (2.0 * 6 * (D // 2) * D)
        # likewise the keys' projection of the token -> image attentions from layer 1 on and of the final one (fold_t2i): 2*S*D*(D/2)
        # + the q.k products 2*S*6*(D/2) become scores = keys x Q'^T (2*S*D*6*heads) + Q' = q W_k (2*6*(D/2)*D)
        t2i_saving = (2.0 * S * D * (D // 2) + 2.0 * S * 6 * (D // 2)) - (2.0 * S * D * HT + 2.0 * 6 * (D // 2) * D)
        # ... and their VALUE projection (l4p_t2i_context): 2*S*D*(D/2) + the P.V products 2*S*6*(D/2) become ctx = probs x keys
        # (2*S*6*heads*D) + the 48 context rows through their head's block of W_v (2*6*D*(D/2)/1): the same count again
        per_qw = TRACK_FLOPS_PER_QUERY_WINDOW - hist_full - cfg.sam_depth * (i2t_projected - i2t_folded) - 2 * cfg.sam_depth * t2i_saving
        shared = 2 * 2.0 * S * D * (D // 2)
        total = n_windows * per_qw * n_queries                       # every window, every query
        total += (n_windows - 1) * 0.5 * hist_full * n_queries       # memory tokens for the windows that have a successor
        total -= shared * (n_queries - 1)                            # first window: shared image-side projections
        total -= (n_windows - 1) * 0.5 * shared * (n_queries - 1)    # later windows: their track-independent temporal half
        gemm += total / n_windows
    return {"gemm": gemm, "conv3d": conv, "attention": attn}


_SHAPE = re.compile(r"^M(\d+) N(\d+) K(\d+) epi(\d) act\d (.*)$")


def executed_flops_of_tag(tag: str, D: int = 1408) -> float:
    """2*M*N*K of one profiled dense launch (tag written by the launcher: "M.. N.. K.. epi.. act.. <form>"), minus the padding the
    kernels add: head dim 88 -> 96 in the QKV projection, per-tap channels 176 -> 192 in the tracker's last up-scaling, patch vector
    1176 -> 1216, the k dimension 48 -> 64 of the folded P x V', and the structural zeros of the block-diagonal token-side weights of
    the folded attentions (head h meets head h's D rows only).  0 for tags that carry no shape (grouped launches)."""
    m = _SHAPE.match(tag)
    if not m:
        return 0.0
    M, N, K, epi, form = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5)
    if epi == 1:
        N = N * 88 // 96
    if epi == 3:
        N = N * 176 // 192
    if K == 1216:
        K = 1176
    if "wgrp" in form and K == 64:
        K = 48
    if N == 8 * D and K == D // 2:
        N = D
    if N == 8 and K == D // 2:
        N = 1
    return 2.0 * M * N * K


def read_prof_detail(lib):
    """[(class, tag, launches, total ms)] of the event profiler's per-tag table."""
    n = lib.l4p_prof_detail(None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    lib.l4p_prof_detail(buf, n)
    rows = []
    for line in buf.value.decode().splitlines():
        cls, tag, cnt, ms = line.split("\t")
        rows.append((cls, tag, int(cnt), float(ms)))
    return rows


def read_prof(lib):
    out = {}
    for c in range(lib.l4p_prof_num_classes()):
        ms, n = C.c_double(0), C.c_longlong(0)
        _lib.check(lib.l4p_prof_read(c, C.byref(ms), C.byref(n)), "l4p_prof_read")
        out[lib.l4p_prof_class_name(c).decode()] = (ms.value, n.value)
    return out


def cpu_baseline(sd, cfg, tasks, batch_cpu, sample_blocks: int = 0, sample_queries: int = 0):
    """Oracle (plain PyTorch fp32 port of the reference algorithm) on the host cores, 1 clip of the SAME workload, a bounded
    sample (~30 s of CPU work) timed IN FULL: patch embed + every encoder block + the dense heads + the tracker on every query
    (``sample_blocks`` / ``sample_queries`` > 0 restrict the encoder blocks / queries timed and scale the rest - not the default).
    Threads: torch's CPU kernels stop scaling far below a 256-core host's count, so two encoder blocks are timed on min(64, cores)
    and on ALL cores first and the measurement runs on the faster setting; both probe times are reported."""
    from oracle import l4p_oracle as orc

    host = os.cpu_count() or 1
    rgb = batch_cpu["rgb_b3thw"]
    dense = [t for t in tasks if t != "track_2d"]
    with torch.no_grad():
        torch.set_num_threads(min(host, 64))
        feats = orc.encoder_forward(sd, rgb, cfg, upto=0)
        x0 = feats[0]
        probe = {}
        for th in sorted({min(host, 64), host}):
            torch.set_num_threads(th)
            orc.encoder_block(sd, "video_encoder.blocks.0.", x0, cfg.heads, cfg.ln_eps)  # (warm the pool)
            t0 = time.time()
            for i in range(2):
                orc.encoder_block(sd, f"video_encoder.blocks.{i}.", x0, cfg.heads, cfg.ln_eps)
            probe[th] = (time.time() - t0) / 2
        threads = min(probe, key=probe.get)
        torch.set_num_threads(threads)
        t0 = time.time()
        feats = orc.encoder_forward(sd, rgb, cfg, upto=0)
        t_embed = time.time() - t0
        t0 = time.time()
        x = feats[0]
        sample_blocks = cfg.depth if sample_blocks <= 0 else min(sample_blocks, cfg.depth)
        for i in range(sample_blocks):
            x = orc.encoder_block(sd, f"video_encoder.blocks.{i}.", x, cfg.heads, cfg.ln_eps)
        t_blocks = (time.time() - t0) / sample_blocks
        fl = [x] * (cfg.depth + 1)  # same shapes/statistics class as the real hooks; values are irrelevant for timing
        om = orc.OracleModel(sd, cfg, use_intrinsics=True)
        t0 = time.time()
        for t in dense:
            om.dense_single(t, fl, batch_cpu["intrinsics_b44t"])
        t_heads = time.time() - t0
        t_track, nq, track_note = 0.0, 0, ""
        if "track_2d" in tasks:
            q = batch_cpu["track_2d_pointquerries_bn3"]
            nq = q.shape[1]
            ns = nq if sample_queries <= 0 else min(sample_queries, nq)
            t0 = time.time()
            orc.track_windowed(sd, cfg, [x], q[:, :ns], batch_cpu["track_2d_pointlabels_bn"][:, :ns], [0])
            t_track = (time.time() - t0) * nq / ns
            track_note = (f" + tracker, all {nq} queries, {t_track:.2f}s" if ns == nq else
                          f" + tracker {ns}/{nq} queries scaled x{nq / ns:g} = {t_track:.2f}s")
    dt = t_embed + t_blocks * cfg.depth + t_heads + t_track
    # "cores" = the threads the port actually ran on (torch intra-op pool); "host_cores" = what the box has (os.cpu_count())
    return {"value": round(16.0 / dt, 4), "unit": "frames/s", "cores": threads, "host_cores": host, "kind": "port",
            "thread_probe_s_per_block": {str(k): round(v, 3) for k, v in probe.items()},
            "sample": (f"1 clip (16x224x224), tasks={'+'.join(tasks)}: oracle (plain PyTorch fp32 port of the reference) on the host CPU, "
                       f"{threads} threads (the faster of {sorted(probe)} on a 2-block probe: {', '.join(f'{k}: {v:.2f}s/block' for k, v in sorted(probe.items()))}); "
                       f"timed patch-embed {t_embed:.2f}s + {sample_blocks}/{cfg.depth} encoder blocks ({t_blocks:.2f}s each{'' if sample_blocks == cfg.depth else f', scaled x{cfg.depth}'}) "
                       f"+ dense heads in full {t_heads:.2f}s{track_note} -> {dt:.1f}s per clip")}


def build_workload(tasks, B, nq, device, rank=0, frames=16, same_data=False, use_intrinsics=None):
    """Model (name-seeded random weights, packed on rank 0 and broadcast once) + one synthetic batch of B clips."""
    cfg = ModelCfg.full()
    model = build_model(os.path.join(ROOT, "configs", "model.yaml"), precision=ENGINE_PRECISION)
    net = model.l4p_model
    net.task_heads = torch.nn.ModuleDict({t: net.task_heads[t] for t in tasks})
    if "camray" in tasks and use_intrinsics is not None:
        # None = configs/model.yaml as shipped (use_intrinsics: false -> the first window estimates K from its ray map,
        # rays_to_intrinsics kernel); True = the Dycheck mode of demo.py:215 (given intrinsics)
        net.task_heads["camray"].use_intrinsics = bool(use_intrinsics)
    sd, pw = None, None
    if rank == 0:
        sd = seeded_state_dict(cfg, tasks=tasks)
        pw = pack_state_dict(sd, cfg, PRECISION[ENGINE_PRECISION][1], device, tasks=tasks)
    pw = broadcast_weights(pw, device)  # RCCL over xGMI, once
    net.set_weights(pw)

    g = torch.Generator().manual_seed(1234 + (0 if same_data else rank))
    rgb = torch.randn([B, 3, frames, 224, 224], generator=g, dtype=torch.float32)
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 224.0
    K[0, 2] = K[1, 2] = 112.0
    batch = {"rgb_b3thw": rgb.to(device), "intrinsics_b44t": K[None, :, :, None].repeat(B, 1, 1, frames).to(device)}
    if "track_2d" in tasks:
        from l4p_amd.data.synthetic import grid_queries

        q = grid_queries(nq)
        batch["track_2d_pointquerries_bn3"] = q.repeat(B, 1, 1).to(device)  # every clip tracks its own nq queries
        batch["track_2d_pointlabels_bn"] = torch.ones(B, nq, device=device)
    return model, batch, sd


def c5_phase_times(net, batch, tasks, rank, world, group):
    """configs[4] once more, piece by piece with a device synchronisation between the pieces (outside the timed region), in the
    order parallel.forward_windows_sharded issues them: 1a = encoder of this rank's windows, x1 = all-gather of the last-layer
    features, 1b = DPT decoders of this rank's windows, x2 = all-gather of the decoded windows, 3 = dense stitching / seam
    alignment / pose chaining (REPLICATED on every rank), T = the tracker recursion over all windows on this rank's query shard
    (in the timed step it runs on its own streams BESIDE 1b / x2; 3 follows both).  On one GPU the same pieces give what one of EIGHT ranks
    would run: an eighth of 1a and 1b, all of 3, the tracker on an eighth of the queries.  All ranks execute this (the exchanges
    are collectives); times are this rank's."""
    from l4p_amd import parallel as par

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) * 1e3

    data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    strides = net.time_strides(data["rgb_b3thw"].shape[2])
    nwin = len(strides)
    B = data["rgb_b3thw"].shape[0]
    dense = [t for t in tasks if t != "track_2d"]
    track = "track_2d" in tasks

    def run_tracker(lasts, r, w):
        d, n = par.shard_track_inputs(data, r, w)
        if n == 0:
            return None
        wins = [par.DecodedWindow(net.cfg.depth, {}, g["last"]) for g in lasts]
        return net.task_heads["track_2d"].forward_windowed(enc_features_bpc_2dlist=wins, time_strides=strides, **d)

    with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
        groups, enc = timed(lambda: par.encode_local_windows(net, data, tasks, rank, world, group))
        res = {"phase1a_encoder_ms": round(enc, 3)}
        lasts, x1 = None, 0.0
        if track:
            lasts, x1 = timed(lambda: par.all_gather_windows(par.local_last_features(groups, B), nwin, rank, world))
        local, dec = timed(lambda: par.decode_encoded_windows(net, data, tasks, groups))
        del groups
        gathered, x2 = timed(lambda: par.all_gather_windows(local, nwin, rank, world))
        windows = [par.DecodedWindow(net.cfg.depth, {k[4:]: v for k, v in g.items() if k.startswith("dec.")}, None) for g in gathered]
        _, st = timed(lambda: net.stitch_windows(windows, data, dense, strides))
        res.update({"phase1b_decoders_ms": round(dec, 3), "exchange_last_ms": round(x1, 3), "exchange_decoded_ms": round(x2, 3),
                    "phase3_dense_ms": round(st, 3), "phase1_ms": round(enc + dec, 3)})
        trk = trk8 = 0.0
        if track:
            _, trk = timed(lambda: run_tracker(lasts, rank, world))
            res["phase3_track_ms"] = round(trk, 3)
            if world == 1:
                _, trk8 = timed(lambda: run_tracker(lasts, 0, 8))
                res["phase3_track_ms_on_an_eighth_of_the_queries"] = round(trk8, 3)
        res["phase3_ms"] = round(st + trk, 3)
        if world == 1 and track:
            # One of EIGHT ranks, measured instead of modelled: rank 0's chunk of windows through the encoder, the tracker on its
            # eighth of the queries over all windows (last-layer features as x1 would deliver them) on its own stream beside
            # the decoders of rank 0's chunk, the join, the replicated stitch of all windows (decoded as x2 would deliver them).
            # Everything a rank does except the three collectives.
            nq8 = par.shard_track_inputs(data, 0, 8)[1]  # (the shard size selects the decoders' stream: parallel.decoder_stream)

            def rank0_of_8():
                g8 = par.encode_local_windows(net, data, tasks, 0, 8, group)
                tr = net.task_heads["track_2d"]
                tr.defer_join = tr.own_stream = True
                tr.start_event = torch.cuda.Event()
                tr.start_event.record(torch.cuda.current_stream())
                ts = par.cu_masked_stream(net.device, os.environ.get("L4P_C5_TRK_CUS"))
                tr.clip_stream_override = [ts] if ts is not None else None
                try:
                    par.decode_encoded_windows_on(par.decoder_stream(net.device, nq8), net, data, tasks, g8)
                    o = run_tracker(lasts, 0, 8)
                    if not par.stitch_beside_tracker():
                        tr.join_streams()
                    net.stitch_windows(windows, data, dense, strides)
                finally:
                    tr.join_streams()
                    tr.defer_join = tr.own_stream = False
                    tr.start_event = None
                    tr.clip_stream_override = None
                return o

            rank0_of_8()
            _, r8 = timed(rank0_of_8)
            res["emulated_rank0_of_8_ms"] = round(r8, 3)
            # The same rank under the SEAM-LOCAL exchange (parallel.stitch_seam_local; L4P_C5_EXCHANGE=seam): its own
            # windows' heads, its own seams against the raw predecessor, the prefix composition of all 30 seam records (taken from an
            # untimed emulation of the 8 ranks), its own frames - instead of the replicated stitch of all 31 windows.
            if par.seam_local_supported(net, dense):
                img_info = tuple(data.get("img_info", net.window_size))
                _, rel_all = par.stitch_seam_local_emulated(net, data, dense, windows, 8)
                s0, e0 = par.window_chunks(nwin, 8)[0]

                def seam_part():
                    st8 = par.SeamLocalState(0, 8, {w: windows[w] for w in range(s0, e0)}, strides, net.window_size[0])
                    k0 = par.seam_phase_k0(net, st8, data, img_info)
                    par.seam_phase_heads(net, st8, data, dense, img_info, k0)
                    par.seam_phase_solve(st8, None, img_info)
                    return par.seam_phase_apply(st8, rel_all)

                def rank0_of_8_seam():
                    g8 = par.encode_local_windows(net, data, tasks, 0, 8, group)
                    tr = net.task_heads["track_2d"]
                    tr.defer_join = tr.own_stream = True
                    tr.start_event = torch.cuda.Event()
                    tr.start_event.record(torch.cuda.current_stream())
                    try:
                        par.decode_encoded_windows_on(par.decoder_stream(net.device, nq8), net, data, tasks, g8)
                        o = run_tracker(lasts, 0, 8)
                        if not par.stitch_beside_tracker():
                            tr.join_streams()
                        seam_part()
                    finally:
                        tr.join_streams()
                        tr.defer_join = tr.own_stream = False
                        tr.start_event = None
                    return o

                _, sp = timed(seam_part)
                _, sp = timed(seam_part)
                rank0_of_8_seam()
                _, r8s = timed(rank0_of_8_seam)
                res["phase3_dense_seam_local_rank0_of_8_ms"] = round(sp, 3)
                res["emulated_rank0_of_8_seam_local_ms"] = round(r8s, 3)
            xb = par.seam_exchange_bytes(B, nwin, 8, tasks=tuple(dense))
            last_bytes = B * 2048 * net.cfg.dim * 4 * (nwin - (par.window_chunks(nwin, 8)[1][1] - par.window_chunks(nwin, 8)[1][0]))
            ds = os.environ.get("L4P_C5_DEC_CUS")
            res["emulated_rank_decoder_cus"] = (ds if ds is not None else ("0,160" if 0 < nq8 <= 16 else "")) or "all"
            res["exchange_bytes_per_rank_of_8"] = {"last_layer_features_all_gather": int(last_bytes),
                                                   "dense_gather_schedule": xb["gather_schedule"],
                                                   "dense_seam_local_schedule": xb["seam_local_schedule"]}
    if world == 1:
        serial = enc + dec + st + trk  # the pieces one after the other on one GPU
        res["serial_sum_ms"] = round(serial, 3)
        # round 3's schedule on 8 ranks: windows sharded, stitch replicated, the tracker (an eighth of the queries) AFTER the decoders
        res["implied_8gpu_ms_tracker_after_decoders"] = round((enc + dec) / 8.0 + st + trk8, 3)
        # this round's: the tracker starts after the encoders and runs beside the decoders; the stitch follows both
        res["implied_8gpu_ms_tracker_beside_decoders"] = round(enc / 8.0 + max(trk8, dec / 8.0) + st, 3)
    return res


def bench_prep(args, rank, world, device, lib):
    """--workload prep: the caller side of the hot path — one decoded 480x854 video of 50 frames per rank,
    resident in HBM as uint8, through l4p_amd.data.prepare_clip (Pillow resize-blur-resize, mirror-pad to 64 frames,
    resize to 224x224, normalise).  HBM-bound byte work: roofline = algorithmic bytes / kernel time vs 8 TB/s."""
    import numpy as np

    from l4p_amd.data import prepare_clip
    from l4p_amd.data.synthetic import synthetic_video

    T, H, W, T_out = 50, 480, 854, 64
    host = synthetic_video(100 + rank, T, H, W)
    frames = torch.from_numpy(host).to(device)

    def step():
        return prepare_clip(frames, (T_out, 224, 224), (224, 224), spacing=0.04)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    lib.l4p_prof_reset()
    lib.l4p_prof_enable(1)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    lib.l4p_prof_enable(0)
    if rank != 0:
        return
    ms, n = read_prof(lib)["preprocess"]
    # algorithmic bytes: every decoded frame read once (uint8) + the network input written once (float32)
    alg = T * H * W * 3 + 3 * T_out * 224 * 224 * 4
    ach = alg * args.steps / (ms * 1e-3) / 1e9
    res = {
        "metric": "video frames/sec prepared (resize-blur-resize, mirror-pad, 224x224 resize, normalise)",
        "value": round(world * T_out * args.steps / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8 (22-bit fixed-point filter) + f32", "data": "synthetic (seeded 480x854 uint8 video)",
        "config": {"workload": f"clip preparation: {T} decoded {H}x{W} frames -> [3,{T_out},224,224] float (one video per GPU per step)"},
        "roofline": {"kernel": "pil_resample_{h,v}_kernel x3 + clip_resize_normalize_kernel<fused last pass>", "bound": "hbm",
                     "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
                     "algorithmic_bytes_per_step": alg, "kernel_ms_per_step": round(ms / args.steps, 4),
                     "launches_per_step": n / args.steps,
                     "method": "algorithmic bytes (frames read once + output written once) / HIP-event-bracketed kernel time"},
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle import preprocess_oracle as po
        ns = 6
        t0 = time.perf_counter()
        po.preprocess_clip(host[:ns], crop_size=(ns, 224, 224), resize_size=(224, 224), spacing=0.04)
        el = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(ns / el, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"oracle/preprocess_oracle.preprocess_clip on {ns} of the {T} frames (numpy, one thread)"}
    print(json.dumps(res))


DEMO_TASKS = ["depth", "flow_2d_backward", "dyn_mask", "track_2d"]  # demo.py:82,99


def bench_demo(args, rank, world, device, lib, selftest):
    """--workload demo: large-N tracking at the demo's scale (625 queries over 7 windows, chunks of 128) next to the dense
    heads of a 64-frame clip; one video per GPU per step (weak scaling)."""
    from l4p_amd.data import prepare_clip
    from l4p_amd.data.synthetic import synthetic_video

    cfg = ModelCfg.full()
    model, _, sd = build_workload(list(DEMO_TASKS), 1, 64, device, rank)
    net = model.l4p_model
    net.task_heads["track_2d"].max_queries = 128   # demo.py:38-40
    net.window_batch = 8                           # as demo/demo.py of this repository (all 7 windows of the clip in one group)
    T_out = 64
    host = synthetic_video(1 + rank, 50, 480, 854)
    clip = prepare_clip(torch.from_numpy(host).to(device), (T_out, 224, 224), (224, 224), spacing=0.04)
    data = {k: (v[None] if torch.is_tensor(v) else v) for k, v in clip.items()}
    nq = int(data["track_2d_pointquerries_bn3"].shape[1])
    nwin = (T_out - 16) // 8 + 1

    def step():
        with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
            return model.forward(data, DEMO_TASKS)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    os.environ["L4P_TRACK_STREAMS"] = os.environ["L4P_HEAD_STREAMS"] = "0"
    lib.l4p_prof_reset()
    lib.l4p_prof_enable(1)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    lib.l4p_prof_enable(0)
    if rank != 0:
        return
    prof = read_prof(lib)
    fl = dict(algorithmic_flops(cfg, DEMO_TASKS, nq, n_windows=nwin))  # per-window average
    # (small / streaming dense products in their own class, FLOPs from the executed shapes: see main)
    fl["gemm_small"] = sum(executed_flops_of_tag(tag, cfg.dim) * cnt for cls, tag, cnt, _ in read_prof_detail(lib) if cls == "gemm_small") / args.steps / nwin
    fl["gemm"] = max(fl["gemm"] - fl["gemm_small"], 0.0)
    classes = {}
    for name, (ms, n) in prof.items():
        if n:
            ent = {"ms_per_step": round(ms / args.steps, 4), "launches_per_step": n / args.steps, "avg_launch_us": round(ms / n * 1e3, 3)}
            if fl.get(name, 0) > 0:
                ent["tflops"] = round(fl[name] * nwin / (ms / args.steps * 1e-3) / 1e12, 2)
            classes[name] = ent
    dom = max((k for k in ("gemm", "conv3d", "attention") if k in classes), key=lambda k: classes[k]["ms_per_step"])
    res = {
        "metric": "frames/sec (depth + flow + motion-seg + 2D/3D tracks), 64-frame 224x224 clip, 625 track queries",
        "value": round(world * T_out * args.steps / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": PRECISION[ENGINE_PRECISION][0], "data": "synthetic (seeded 480x854 video, name-seeded random weights)",
        "rccl_ranks": selftest["ranks"] if selftest.get("backend") == "nccl" else (1 if world == 1 else 0),
        "config": {"workload": f"demo generic video (demo/demo.py:84-100): 1 clip of {T_out} frames = {nwin} windows per GPU, tasks "
                               f"{'+'.join(DEMO_TASKS)}, {nq} grid queries in chunks of {net.task_heads['track_2d'].max_queries}, windows batched {net.window_batch} at a time",
                   "queries": nq, "windows": nwin, "tasks": DEMO_TASKS},
        "roofline": {"kernel": {"gemm": "gemm8p_kernel<0> / gemm_kernel<bf16,MODE0>", "conv3d": "gemm8p_kernel<1> / gemm_kernel<bf16,MODE1>",
                                "attention": "attn64_kernel<bf16,88> (encoder attention, one wave per SIMD, 64 query rows per wave; attn_kernel<bf16,96,64> below 256 tiles)"}[dom], "bound": "mfma", "achieved": classes[dom]["tflops"],
                     "peak": PRECISION[ENGINE_PRECISION][2] / 1e12, "unit": "TFLOP/s", "frac": round(classes[dom]["tflops"] / (PRECISION[ENGINE_PRECISION][2] / 1e12), 4),
                     "traffic": None, "method": "algorithmic FLOPs / HIP-event-bracketed kernel time, second pass of the same K steps",
                     "algorithmic_flops_per_step": fl[dom] * nwin},
        "kernel_classes": classes,
    }
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c5", "prep", "demo"])
    ap.add_argument("--frames", type=int, default=256, help="c5: length of the long video")
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU per step (default: 1 for c2; c3: 4 on one GPU = configs[2], "
                                                         "8 per GPU on N > 1 GPUs = configs[3]: batch 64 over 8 GPUs)")
    ap.add_argument("--use-intrinsics", action="store_true", help="camray head with given intrinsics (demo.py:215) instead of "
                                                                  "the shipped use_intrinsics=false (K estimated from the ray map)")
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--group", type=int, default=16, help="c5: windows batched through encoder + dense decoders per launch group (16: whole rounds of 256x256 tiles in the encoder linears; 4 -> 16: +6.8 % on one GPU)")
    ap.add_argument("--precision", default="bf16", choices=sorted(PRECISION), help="engine dtype: bf16 (BASELINE.json's configs; default), "
                    "16-mixed (IEEE half: the reference demo's own mode), 32-true (exact-f32 parity engine)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-quick", action="store_true",
                    help="contract tests only: time 4 encoder blocks and 8 queries and scale (the default times the whole clip)")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--host-io", action="store_true", help="c2 / c3: after the timed steps, K more with the inputs coming from and every output going to "
                                                           "pinned host memory; reported as pcie_inclusive (never as value)")
    args = ap.parse_args()
    global ENGINE_PRECISION
    ENGINE_PRECISION = args.precision

    rank, world, local = init_distributed()
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    lib = _lib.load()
    # N > 1: the collectives of the path (weight broadcast, unequal-chunk window all-gather, MAX all-reduce) on the live
    # backend, before anything is timed; its verdict travels in the JSON line
    selftest = collective_selftest(device) if world > 1 else {"ranks": 1, "backend": None, "ok": True}
    if args.workload == "prep":
        return bench_prep(args, rank, world, device, lib)
    if args.workload == "demo":
        return bench_demo(args, rank, world, device, lib, selftest)

    cfg = ModelCfg.full()
    tasks = ["depth"] if args.workload == "c2" else list(ALL_TASKS)
    B = args.batch or ((4 if world == 1 else 8) if args.workload == "c3" else 1)
    c5 = args.workload == "c5"  # configs[4]: ONE long video, its windows sharded over the ranks (strong scaling)

    model, batch, sd = build_workload(tasks, B, args.queries, device, rank, frames=args.frames if c5 else 16, same_data=c5,
                                      use_intrinsics=True if args.use_intrinsics else None)
    if c5:
        model.l4p_model.always_use_windowed_version = True

    def step():
        # the model mirrors the reference's stdout messages ("Joint alignment is not possible ..." for a depth-only task
        # list, l4p_videomae.py:322); stdout of this script carries the ONE JSON line only
        with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
            if c5:
                from l4p_amd.parallel import forward_windows_sharded
                return forward_windows_sharded(model.l4p_model, batch, tasks, rank, world, group=args.group)
            return model.forward(batch, tasks)

    for _ in range(args.warmup):
        step()
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, no instrumentation ----
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # ---- kernel-class durations: the same K steps again with every launch bracketed by HIP events on the launch
    #      stream.  Kept out of the timed region because the event packets themselves cost ~13 % of a step at batch 1
    #      (15.1 ms -> 17.5 ms measured); kernel durations are unaffected up to a few %.
    if not args.no_prof and rank == 0:
        # kernel durations are taken with the clips' trackers and the side-stream decoders serialised on one stream
        saved = {k: os.environ.get(k) for k in ("L4P_TRACK_STREAMS", "L4P_HEAD_STREAMS")}  # (an A/B run may have set them)
        os.environ["L4P_TRACK_STREAMS"] = os.environ["L4P_HEAD_STREAMS"] = "0"
        lib.l4p_prof_reset()
        lib.l4p_prof_enable(1)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        lib.l4p_prof_enable(0)
        for k, v in saved.items():  # the --host-io pass below runs in the configuration `value` was measured in
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if world > 1:
        dist.barrier()
    # ---- the same K steps once more with the boundary handed HOST buffers (what a DataLoader delivers, l4p.py:54-66 moves them to
    #      the device): inputs start in pinned host memory, every output tensor ends in pinned host memory.  Reported beside
    #      `value` as "pcie_inclusive"; never `value` itself.
    pcie = None
    if not c5 and args.host_io:
        hbatch = {k: (v.cpu().pin_memory() if torch.is_tensor(v) else v) for k, v in batch.items()}
        hout = {}

        def host_step():
            with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
                out = model.forward({k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in hbatch.items()}, tasks)
            for k, v in out.items():
                if torch.is_tensor(v):
                    if k not in hout:
                        hout[k] = torch.empty(v.shape, dtype=v.dtype).pin_memory()
                    hout[k].copy_(v, non_blocking=True)

        host_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            host_step()
        torch.cuda.synchronize()
        dth = time.perf_counter() - t1
        pcie = {"value": round(world * B * 16 * args.steps / dth, 3), "unit": "frames/s (this rank's rate x ranks)", "ms_per_step": round(dth / args.steps * 1e3, 3),
                "h2d_bytes_per_step": int(sum(v.numel() * v.element_size() for v in hbatch.values() if torch.is_tensor(v))),
                "d2h_bytes_per_step": int(sum(v.numel() * v.element_size() for v in hout.values())),
                "note": "inputs from pinned host memory, all outputs copied to pinned host memory, inside the timed steps"}
        del hbatch, hout
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    phases = c5_phase_times(model.l4p_model, batch, tasks, rank, world, args.group) if c5 else None

    if rank != 0:
        return
    frames = (B * args.frames if c5 else world * B * 16) * args.steps
    res = {
        "metric": "frames/sec (all heads), 16x224x224 clip; encoder MFMA-roofline %" if args.workload in ("c3", "c5") else
                  "frames/sec (depth head only), 16x224x224 clip; encoder MFMA-roofline %",
        "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if c5 else "weak", "vs_baseline": None,
        "dtype": PRECISION[ENGINE_PRECISION][0], "data": "synthetic (randn clips, name-seeded random weights of the VideoMAE-v2-giant + DPT geometry)",
        "rccl_ranks": selftest["ranks"] if selftest.get("backend") == "nccl" else (1 if world == 1 else 0),
        "rccl_selftest": None if world == 1 else selftest,
        "config": {"workload": (f"configs[1]: single MI355X, depth head only, {PRECISION[ENGINE_PRECISION][0]}, batch=1 16-frame 224x224 clip" if args.workload == "c2"
                                 else f"configs[4]: one {args.frames}-frame video -> {(args.frames - 16) // 8 + 1} overlapping 16-frame windows sharded over the ranks, all heads, on-GPU pose / window alignment, {args.queries} track queries" if c5
                                 else (f"configs[2]: single MI355X, all heads (depth+flow+track2d/3d+motion-seg+pose), {PRECISION[ENGINE_PRECISION][0]}, batch={B} clips, {args.queries} track queries per clip" if world == 1
                                       else f"configs[3]: {world}xMI355X data-parallel over clips, all heads, {PRECISION[ENGINE_PRECISION][0]}, batch={world * B} clips ({B} per GPU), {args.queries} track queries per clip, RCCL weight broadcast")),
                   "clips_per_gpu_per_step": B, "tasks": tasks,
                   "camray_use_intrinsics": bool(args.use_intrinsics) if "camray" in tasks else None,
                   "parallelism": (f"windows sharded over {world} rank(s); all-gather of the last-layer features after the encoders, query-sharded tracker beside the decoders, all-gather of the decoded windows, stitching replicated" if c5
                                   else f"dp{world} (clips sharded, no collective in the step)")},
    }
    if pcie:
        res["pcie_inclusive"] = pcie
    if phases:
        res.update(phases)
        if world == 1 and "implied_8gpu_ms_tracker_beside_decoders" in phases:
            # strong scaling against the MEASURED one-GPU step (in which the tracker already runs beside the decoders); the
            # exchanges (11.5 MB of last-layer features and ~13 MB of decoded outputs per window) are not priced: no N > 1 lease
            t1 = dt / args.steps * 1e3
            res["implied_8gpu_speedup_tracker_after_decoders"] = round(t1 / phases["implied_8gpu_ms_tracker_after_decoders"], 3)
            res["implied_8gpu_speedup_tracker_beside_decoders"] = round(t1 / phases["implied_8gpu_ms_tracker_beside_decoders"], 3)
            if "emulated_rank0_of_8_ms" in phases:
                res["implied_8gpu_speedup_emulated_rank"] = round(t1 / phases["emulated_rank0_of_8_ms"], 3)
            if "emulated_rank0_of_8_seam_local_ms" in phases:
                res["implied_8gpu_speedup_emulated_rank_seam_local"] = round(t1 / phases["emulated_rank0_of_8_seam_local_ms"], 3)
    if not args.no_prof:
        prof = read_prof(lib)
        nwin_total = (args.frames - 16) // 8 + 1 if c5 else 1
        fl = algorithmic_flops(cfg, tasks, args.queries if "track_2d" in tasks else 0, n_windows=nwin_total)
        nwin_rank0 = 1
        if c5:  # rank 0's share: its chunk of windows (the tracker term is approximate: queries, not windows, are sharded)
            from l4p_amd.parallel import window_chunks
            s0, e0 = window_chunks((args.frames - 16) // 8 + 1, world)[0]
            nwin_rank0 = e0 - s0
        # The dense products that are latency- or HBM-bound by construction (fewer than 1024 rows: the tracker's token side; the
        # tracker's folded cross-attention products: [N*P, 1408] x [1408, 48] scores, 48-term delta, softmax x keys) are profiled in
        # their own class, "gemm_small".  Their FLOPs are taken from the executed shapes (the FLOP model counts them inside "gemm";
        # tests/test_bench_flops_cpu.py holds the model to the executed total): what is left prices the MFMA-bound GEMM class.
        small_exec = sum(executed_flops_of_tag(tag, cfg.dim) * cnt for cls, tag, cnt, _ in read_prof_detail(lib) if cls == "gemm_small") / args.steps
        fl = dict(fl)
        fl["gemm_small"] = small_exec / (B * nwin_rank0)
        fl["gemm"] = max(fl["gemm"] - fl["gemm_small"], 0.0)
        classes = {}
        for name, (ms, n) in prof.items():
            if n == 0:
                continue
            per_step_ms = ms / args.steps
            ent = {"ms_per_step": round(per_step_ms, 4), "launches_per_step": n / args.steps,
                   "avg_launch_us": round(ms / n * 1e3, 3)}
            if name in fl and fl[name] > 0:
                ent["tflops"] = round(fl[name] * B * nwin_rank0 / (per_step_ms * 1e-3) / 1e12, 2)
            classes[name] = ent
        mfma = [k for k in ("gemm", "conv3d", "attention") if k in classes]
        dom = max(mfma, key=lambda k: classes[k]["ms_per_step"])
        kern = {"gemm": "gemm8p_kernel<0> / gemm_kernel<bf16,MODE0> (linear / 1x1x1 conv / ConvTranspose GEMM; >= 1024 rows, one weight matrix)",
                "conv3d": "conv3_halo_kernel (LDS-halo 3x3x3 conv) / gemm_kernel<bf16,MODE1> (implicit-GEMM 3x3x3 conv, low-resolution levels)",
                "attention": "attn64_kernel<bf16,88> (encoder attention, one wave per SIMD, 64 query rows per wave; attn_kernel<bf16,96,64> below 256 tiles)"}

        # HBM bytes per launch of the class from the committed PMC passes (profiles/r01_c3_hbm_traffic.*: rocprofv3
        # FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate passes); PMC cannot be collected from inside this
        # process, so the figure is the one measured on the same command and is only attached to the workload it was taken on
        traffic = {}
        import glob

        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_hbm_traffic.json")))
        tpath = cands[-1] if cands else ""
        traffic_note = "no PMC traffic file for this workload"
        if args.workload == "c3" and B == 4 and tpath:
            with open(tpath) as f:
                tj = json.load(f)
            # only figures measured on THESE kernels: the file carries the hash of the kernel sources it was taken on
            if ENGINE_PRECISION != "bf16":
                traffic_note = f"profiles/{os.path.basename(tpath)} was measured on the bf16 engine: not attached to a {PRECISION[ENGINE_PRECISION][0]} line"
            elif tj.get("kernel_tree") == _lib.kernel_tree_hash():
                traffic = {k: v["hbm_total"] for k, v in tj["per_class_bytes_per_launch"].items()}
                traffic_note = f"profiles/{os.path.basename(tpath)} (kernel tree {tj['kernel_tree']})"
            else:
                traffic_note = (f"profiles/{os.path.basename(tpath)} was measured on kernel tree {tj.get('kernel_tree')}, this is "
                                f"{_lib.kernel_tree_hash()}: not attached (re-run tools/make_profiles.sh)")

        def roof(k):
            a = classes[k]["tflops"]
            peak = PRECISION[ENGINE_PRECISION][2] / 1e12
            return {"kernel": kern[k], "bound": "mfma", "achieved": a, "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(a / peak, 4), "traffic": traffic.get(k),
                    "traffic_unit": f"HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of this command): {traffic_note}",
                    "method": "algorithmic FLOPs / HIP-event-bracketed kernel time, second pass of the same K steps",
                    "avg_launch_us": classes[k]["avg_launch_us"], "launches_per_step": classes[k]["launches_per_step"],
                    "algorithmic_flops_per_step": fl[k] * B * nwin_rank0}

        res["roofline"] = roof(dom)
        for k in ("attention", "gemm", "conv3d"):  # (the north star names the attention; the GEMM class is what earlier rounds reported)
            if k in classes and dom != k and "tflops" in classes[k]:
                res["roofline_" + k] = roof(k)
        res["kernel_classes"] = classes
    if world == 1 and not args.no_cpu_baseline:
        bc = {k: (v[:1].cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
        if c5:  # one 16-frame window of the long video (the CPU port is timed per clip of 16 frames)
            bc = {k: (v[..., :16, :, :] if k == "rgb_b3thw" else v[..., :16] if k == "intrinsics_b44t" else v) for k, v in bc.items()}
        with contextlib.redirect_stdout(sys.stderr):
            res["cpu_baseline"] = (cpu_baseline(sd, cfg, tasks, bc, sample_blocks=4, sample_queries=8) if args.cpu_baseline_quick else
                                   cpu_baseline(sd, cfg, tasks, bc))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
