"""One process per GPU.  The only collective on the path is the one-off broadcast of the packed weight
arena from rank 0 (RCCL over xGMI when the backend is "nccl"); clips/windows are then sharded with no
collective in the step."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .packing import PackedWeights


def _collectives_on(world: Optional[int] = None) -> bool:
    """True when the exchange steps must go through torch.distributed.  ``world`` is the number of ranks the CALLER shards
    over (None: the whole process group).  A caller that works on its own (``world == 1``: L4P_VideoMAE.forward with
    ``window_batch > 1`` inside a torchrun job of data-parallel clips) never enters a collective, whatever the size of the
    process group; a one-rank GROUP skips them unless L4P_FORCE_COLLECTIVES=1: the single-GPU boxes of the test pool then still
    push the very same tensors (packed arena, decoded windows, query shards) through RCCL's broadcast / all-gather with one
    rank (tests/test_rccl_one_rank_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    gsize = dist.get_world_size()
    if world is None:
        world = gsize
    if world > 1:
        if world != gsize:
            raise ValueError(f"sharding over {world} ranks inside a process group of {gsize}: pass the group's size")
        return True
    return gsize == 1 and os.environ.get("L4P_FORCE_COLLECTIVES", "0") == "1"


def env_rank() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank()
    forced = os.environ.get("L4P_FORCE_COLLECTIVES", "0") == "1"
    if (world > 1 or forced) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # these hosts support dmabuf IPC only (RCCL needs it)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_weights(weights: Optional[PackedWeights], device: torch.device, src: int = 0) -> PackedWeights:
    """Rank ``src`` holds the packed arena; every other rank receives layout + bytes.  One collective."""
    if not _collectives_on():
        assert weights is not None
        return weights
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        assert weights is not None
        meta = [(weights.layout, int(weights.arena.numel()), weights.meta)]
    dist.broadcast_object_list(meta, src=src)
    layout, nbytes, m = meta[0]
    if rank != src:
        weights = PackedWeights.empty_like_layout(layout, nbytes, device, m)
    dist.broadcast(weights.arena, src=src)
    return weights


def shard(n_items: int, rank: int, world: int) -> List[int]:
    """Static round-robin assignment of independent clips / windows to ranks."""
    return list(range(rank, n_items, world))


# ---------------------------------------------------------------------------------------------------------------------
# Config 5: one long video -> overlapping 16-frame windows sharded over the ranks.
# The expensive, window-local work (encoder + DPT decoders) runs on the rank that owns the window; two exchange steps (all-gather
# of the last-layer features right after the encoders, of the decoded windows after the decoders); the cheap sequential part
# (overlap alignment, stitching, pose chaining) then runs replicated on every rank from identical inputs, and the tracker - a
# recursion over windows that is independent per query - runs on a shard of the queries, beside the decoders, and its results
# are all-gathered.  The stitched outputs are therefore bit-identical to the single-GPU windowed forward.
# ---------------------------------------------------------------------------------------------------------------------
def window_chunks(n_windows: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, end) window ranges per rank, sizes differing by at most one (31 windows, 8 ranks -> 4,4,4,4,4,4,4,3)."""
    q, r = divmod(n_windows, world)
    out, s = [], 0
    for k in range(world):
        e = s + q + (1 if k < r else 0)
        out.append((s, e))
        s = e
    return out


class DecodedWindow:
    """Stand-in for EncoderFeatures of a window whose heavy work is already done: ``decoded[task]`` is the DPT decoder
    output of that task (dense_heads._decode returns it as is), ``last`` the float last-layer features the tracker reads."""

    def __init__(self, depth: int, decoded: dict, last: Optional[torch.Tensor]):
        self.depth, self.decoded, self.last = depth, decoded, last

    def __len__(self) -> int:
        return self.depth + 1

    def f32(self, i: int) -> torch.Tensor:
        if i not in (-1, self.depth) or self.last is None:
            raise KeyError(f"decoded window keeps only the last-layer features (asked for {i})")
        return self.last

    def __getitem__(self, i: int) -> torch.Tensor:
        return self.f32(i)


def all_gather_windows(local: dict, n_windows: int, rank: int, world: int) -> List[dict]:
    """local: {window id: {key: tensor}} for this rank's chunk -> list over ALL windows of {key: tensor}.
    One all_gather per key on a [chunk_max, ...] block (chunks differ by at most one window; the pad slot is ignored)."""
    chunks = window_chunks(n_windows, world)
    if not _collectives_on(world):
        return [local[w] for w in range(n_windows)]
    cmax = max(e - s for s, e in chunks)
    s0, e0 = chunks[rank]
    keys = sorted(local[s0].keys()) if e0 > s0 else None
    meta = [None] * world
    dist.all_gather_object(meta, None if keys is None else {k: (tuple(local[s0][k].shape), local[s0][k].dtype) for k in keys})
    ref = next(m for m in meta if m is not None)
    dev = next(iter(local[s0].values())).device if keys else None
    if dev is None:  # a rank without windows (more ranks than windows) still has to take part in the collectives
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    out: List[dict] = [dict() for _ in range(n_windows)]
    for k in sorted(ref.keys()):
        shape, dtype = ref[k]
        block = torch.zeros((cmax,) + shape, dtype=dtype, device=dev)
        for j, w in enumerate(range(s0, e0)):
            block[j].copy_(local[w][k])
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(parts, block)
        for r, (s, e) in enumerate(chunks):
            for j, w in enumerate(range(s, e)):
                out[w][k] = parts[r][j]
    return out


def shard_queries(n_queries: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) of the track queries owned by ``rank``."""
    return window_chunks(n_queries, world)[rank]


def all_gather_queries(x: torch.Tensor, n_queries: int, rank: int, world: int, dim: int = 1) -> torch.Tensor:
    """Inverse of shard_queries along ``dim`` (shards differ by at most one query: padded to the largest)."""
    if not _collectives_on(world):
        return x
    chunks = window_chunks(n_queries, world)
    cmax = max(e - s for s, e in chunks)
    xm = x.movedim(dim, 0).contiguous()
    block = torch.zeros((cmax,) + tuple(xm.shape[1:]), dtype=x.dtype, device=x.device)
    block[: xm.shape[0]].copy_(xm)
    parts = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(parts, block)
    full = torch.cat([parts[r][: e - s] for r, (s, e) in enumerate(chunks)], dim=0)
    return full.movedim(0, dim).contiguous()


def encode_local_windows(net, data: dict, tasks: List[str], rank: int, world: int, group: int = 1) -> list:
    """Phase 1a (sharded): the encoder of this rank's windows, ``group`` windows per launch group.
    Returns [(window ids, EncoderFeatures of that group)] - the hook features the DPT decoders read stay on the device until
    decode_encoded_windows has run (23 MB per window in bf16), the float last-layer features are what the tracker reads."""
    rgb = data["rgb_b3thw"]
    ws = net.window_size[0]
    strides = net.time_strides(rgb.shape[2])
    s0, e0 = window_chunks(len(strides), world)[rank]
    tf, tT = net._taps(tasks)
    groups = []
    # group > 1: that many windows ride through the encoder / decoders as one batch (better filled GEMM tiles: +20 % at
    # group 4).  The batch size selects kernels (split-K of the small convs, the KV split of attention), so the result then
    # equals the window-by-window one only up to summation order (1e-5 relative in f32 mode); group = 1 is bit-identical.
    for g0 in range(s0, e0, group):
        ws_ids = list(range(g0, min(g0 + group, e0)))
        clip = torch.cat([rgb[:, :, int(strides[w]):int(strides[w]) + ws] for w in ws_ids], dim=0)
        groups.append((ws_ids, net.video_encoder(clip, tf, tT)))
    return groups


def local_last_features(groups: list, batch: int) -> dict:
    """{window id: {"last": float last-layer features [B, P, C]}} of the encoded groups (what the tracker's exchange moves)."""
    local = {}
    for ws_ids, feats in groups:
        last = feats.f32(-1)
        for j, w in enumerate(ws_ids):
            local[w] = {"last": last[j * batch:(j + 1) * batch].contiguous()}
    return local


_MASKED_STREAMS: dict = {}


def cu_masked_stream(device: torch.device, spec: Optional[str]):
    """A stream of ``device`` restricted to the CUs ``"first,count"`` (l4p_stream_create_cu_mask), cached; None for an empty spec.
    Used by the sharded long-video path to keep a slice of the chip free of the decoders' long-running workgroups: a tracker kernel
    of a rank's query shard is small, but it can only start on a CU with free registers - and a 3x3x3-conv workgroup holds a CU's
    whole register file for a millisecond (measured on one GPU: decoders 27 ms + tracker 49 ms side by side = 76 ms, no overlap)."""
    if not spec or device.type != "cuda":
        return None
    key = (device.index, spec)
    if key not in _MASKED_STREAMS:
        import ctypes as C

        from . import _lib

        first, count = (int(x) for x in spec.split(","))
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.load().l4p_stream_create_cu_mask(first, count, C.byref(h)), "l4p_stream_create_cu_mask")
        _MASKED_STREAMS[key] = torch.cuda.ExternalStream(h.value, device=device)
    return _MASKED_STREAMS[key]


def stitch_beside_tracker() -> bool:
    """The dense stitch / seam alignment of the sharded long video is queued BEFORE the tracker's streams are joined: it runs beside the
    tail of the recursion (default since round 6: one GPU 540 -> 528 ms per 256-frame video, a rank of eight with the seam-local exchange
    79.5 -> 76.3 ms; L4P_TRACK_BESIDE_STITCH=0 joins first).  See forward_windows_sharded for the history of this switch."""
    return os.environ.get("L4P_TRACK_BESIDE_STITCH", "1") != "0"


def decoder_stream(device: torch.device, nq_local: int):
    """The stream the DPT decoders of a rank's windows are queued on while the tracker of its query shard runs beside them.
    L4P_C5_DEC_CUS="first,count" confines them to those CUs ("" / "0": the whole chip, the main stream).  Default: 160 of the 256 CUs when
    the shard is small (<= 16 queries).  Measured on one of eight ranks of configs[4] (tools/probes/c5_rank_timeline.py, c5_rank_masks.sh):
    beside chip-filling conv kernels every small tracker kernel waits for a round of conv workgroups to end (~90 us each: 4 windows of the
    recursion advance in the 28 ms the decoders take, the pieces simply add up: 89.4 ms); with 96 CUs kept free the tracker's chain runs
    on unhindered while the decoders take 1.4x as long underneath it: 81 ms (160 - 176 CUs alike; 224: 87.6, 128: 83.4).  With 64
    queries on one GPU the tracker's own kernels fill the chip and the mask only costs: off."""
    spec = os.environ.get("L4P_C5_DEC_CUS")
    if spec is None:
        spec = "0,160" if 0 < nq_local <= 16 else ""
    if spec in ("", "0"):
        return None
    return cu_masked_stream(device, spec)


def decode_encoded_windows_on(stream, net, data: dict, tasks: List[str], groups: list) -> dict:
    """decode_encoded_windows queued on ``stream`` (after everything queued on the current stream so far); the current stream waits
    for it again before it returns (its host work is done then, its kernels are not), and the results are marked as used there."""
    if stream is None:
        return decode_encoded_windows(net, data, tasks, groups)
    main = torch.cuda.current_stream()
    stream.wait_stream(main)
    with torch.cuda.stream(stream):
        local = decode_encoded_windows(net, data, tasks, groups)
    main.wait_stream(stream)
    for d in local.values():
        for v in d.values():
            v.record_stream(main)
    return local


def decode_encoded_windows_async(stream, net, data: dict, tasks: List[str], groups: list, after):
    """decode_encoded_windows queued on ``stream`` behind the event ``after`` ONLY (the encoder of these windows) - not behind what the
    current stream has queued since: the all-gather of the last-layer features runs beside the decoders.  Returns (results, join):
    ``join()`` makes the then-current stream wait for the decoders and hands it the results; the caller keeps ``groups`` alive until
    then (their blocks were allocated on the launching stream, which does not wait for ``stream`` before that)."""
    stream.wait_event(after)
    with torch.cuda.stream(stream):
        local = decode_encoded_windows(net, data, tasks, groups)

    def join() -> None:
        cur = torch.cuda.current_stream()
        cur.wait_stream(stream)
        for d in local.values():
            for v in d.values():
                v.record_stream(cur)

    return local, join


def decode_encoded_windows(net, data: dict, tasks: List[str], groups: list) -> dict:
    """Phase 1b (sharded): the DPT decoders of the encoded groups -> {window id: {"dec.<task>": tensor}}."""
    img_info = tuple(data.get("img_info", net.window_size))
    dense = [t for t in tasks if t != "track_2d"]
    B = data["rgb_b3thw"].shape[0]
    local = {}
    for ws_ids, feats in groups:
        dec = {t: net.task_heads[t]._decode(feats, img_info) for t in dense}
        for j, w in enumerate(ws_ids):
            local[w] = {"dec." + t: dec[t][j * B:(j + 1) * B].contiguous() for t in dense}
    return local


def decode_local_windows(net, data: dict, tasks: List[str], rank: int, world: int, group: int = 1) -> dict:
    """Phase 1 in one piece (encoder + DPT decoders of this rank's windows) -> {window id: {key: tensor}} with the decoded
    dense outputs and, when the tracker is asked for, the last-layer features.  forward_windows_sharded runs the two halves
    apart (the tracker starts between them); this form serves the phase-by-phase measurements and the emulated-rank tests."""
    local = {}
    B = data["rgb_b3thw"].shape[0]
    # (group by group, so that only one group's hook features are alive at a time)
    rgb = data["rgb_b3thw"]
    strides = net.time_strides(rgb.shape[2])
    s0, e0 = window_chunks(len(strides), world)[rank]
    ws = net.window_size[0]
    tf, tT = net._taps(tasks)
    for g0 in range(s0, e0, group):
        ws_ids = list(range(g0, min(g0 + group, e0)))
        clip = torch.cat([rgb[:, :, int(strides[w]):int(strides[w]) + ws] for w in ws_ids], dim=0)
        grp = [(ws_ids, net.video_encoder(clip, tf, tT))]
        local.update(decode_encoded_windows(net, data, tasks, grp))
        if "track_2d" in tasks:
            for w, item in local_last_features(grp, B).items():
                local[w].update(item)
    return local


def stitch_gathered_windows(net, data: dict, tasks: List[str], gathered: List[dict], rank: int, world: int) -> dict:
    """Phase 3: replicated stitching / alignment from the gathered windows; the tracker runs on this rank's query shard
    (its outputs hold that shard only - all_gather_queries puts them back together)."""
    strides = net.time_strides(data["rgb_b3thw"].shape[2])
    windows = [DecodedWindow(net.cfg.depth, {k[4:]: v for k, v in g.items() if k.startswith("dec.")}, g.get("last"))
               for g in gathered]
    d = dict(data)
    local_tasks = list(tasks)
    if "track_2d" in tasks:
        d, nq_local = shard_track_inputs(data, rank, world)
        if nq_local == 0:
            local_tasks.remove("track_2d")
    return net.stitch_windows(windows, d, local_tasks, strides)


def shard_track_inputs(data: dict, rank: int, world: int) -> Tuple[dict, int]:
    """The caller's batch with the track queries cut down to this rank's shard; second value: the shard's size."""
    d = dict(data)
    nq = data["track_2d_pointquerries_bn3"].shape[1]
    q0, q1 = shard_queries(nq, rank, world)
    d["track_2d_pointquerries_bn3"] = data["track_2d_pointquerries_bn3"][:, q0:q1].contiguous()
    d["track_2d_pointlabels_bn"] = data["track_2d_pointlabels_bn"][:, q0:q1].contiguous()
    return d, q1 - q0


def forward_windows_sharded(net, data: dict, tasks: List[str], rank: Optional[int] = None, world: Optional[int] = None,
                            group: int = 1) -> dict:
    """L4P_VideoMAE.forward for a long clip with its windows sharded over the ranks (see the block comment above).
    ``net``: l4p_amd.models.l4p_videomae.L4P_VideoMAE with weights set on every rank (broadcast_weights).

    Order of the work on a rank (round 4):
      1a  encoder of its windows                                         (sharded)
      x1  all-gather of the float last-layer features (11.5 MB per window and clip) - only when the tracker is asked for
      1b  DPT decoders of its windows, queued on the main stream         (sharded)
      T   the tracker recursion over ALL windows on this rank's query shard, queued on the tracker's own streams, which wait for
          x1 only: a chain of ~31 x 130 small dependent launches (host-issue-bound at 8 queries per rank) that runs BESIDE 1b
      x2  all-gather of the decoded dense windows (L4P_C5_EXCHANGE=seam: the seam-local exchange below instead - K broadcast, one
          tail per chunk boundary, 18 floats per seam; steps 3 then runs on the rank's own windows only)
      J   join the tracker streams
      3   dense stitching / seam alignment / pose chaining               (replicated, identical inputs on every rank)
      x3  all-gather of the query shards
    The arithmetic is that of decode_local_windows + stitch_gathered_windows (the emulated-rank tests compare against those);
    only the order in which independent work is issued differs: the tracker no longer waits for the decoders.
    3 is queued before J (the seam alignment beside the still-running recursion; L4P_TRACK_BESIDE_STITCH=0: J first).
    Round 4 saw the alignment's pointmap kernel compute zeros in that schedule and kept the two apart; round 5 found the cause - a
    gfx950 interaction between MFMAs of one wave and packed-FP32 instructions with a swizzled src1 of another wave on the same SIMD
    (csrc/common.hpp, tools/check_isa.py) - and removed the affected instruction form from the library; both orders are tested
    (tests/test_stream_overlap_gpu.py), and round 6 made this one the default after a soak (tools/probes/soak_beside_stitch.sh: the
    emulated-rank tests four times over, twelve full-size 256-frame forwards bit-identical)."""
    if rank is None or world is None:
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in data.items()}
    T = data["rgb_b3thw"].shape[2]
    assert T % net.window_stride_T == 0 and T >= net.window_size[0]
    strides = net.time_strides(T)
    nwin = len(strides)
    B = data["rgb_b3thw"].shape[0]
    track = "track_2d" in tasks
    dense = [t for t in tasks if t != "track_2d"]
    groups = encode_local_windows(net, data, tasks, rank, world, group)
    trk, trk_out = None, None
    try:
        lasts, ready = None, None
        d_trk, nq_local = None, 0
        # Decoders confined to a part of the chip (a small query shard, decoder_stream) are queued NOW, behind the encoder only: they run
        # beside the all-gather of the last-layer features (311 MB received per rank of eight over xGMI, during which nothing else
        # computes) as well as beside the tracker.  On the whole chip (no mask) they stay behind the exchange, as before.
        local, dec_join = None, None
        if track and dense and net.device.type == "cuda":
            nq_pre = shard_queries(data["track_2d_pointquerries_bn3"].shape[1], rank, world)
            early_stream = decoder_stream(net.device, nq_pre[1] - nq_pre[0])
            if early_stream is not None and os.environ.get("L4P_C5_DEC_EARLY", "1") != "0":
                enc_done = torch.cuda.Event()
                enc_done.record(torch.cuda.current_stream())
                local, dec_join = decode_encoded_windows_async(early_stream, net, data, tasks, groups, enc_done)
        if track:
            lasts = all_gather_windows(local_last_features(groups, B), nwin, rank, world)
            # the query shard is cut BEFORE `ready` is recorded: for B > 1 the [:, q0:q1] slices are copies, and their copy kernels
            # must sit in front of the event the tracker's streams wait for - not behind the decoders queued below, where the
            # tracker would read the shard before it exists (round-4 advisor finding)
            d_trk, nq_local = shard_track_inputs(data, rank, world)
            if net.device.type == "cuda":
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream())
        # The decoders are QUEUED first and the tracker second: issuing the recursion takes the host ~1.7 ms per window (130
        # launches) whatever the query count - at 8 queries per rank that is all it costs - so the GPU works through the decoders
        # while the host is still feeding the tracker's stream.  The tracker's streams wait for `ready` (the gathered features),
        # not for the decoders queued behind it.
        # (L4P_C5_DEC_CUS / L4P_C5_TRK_CUS = "first,count": CU-masked streams for the decoders / the tracker, see cu_masked_stream)
        if dense and dec_join is None:
            dec_stream = decoder_stream(net.device, nq_local) if track else None
            local = decode_encoded_windows_on(dec_stream, net, data, tasks, groups)
            del groups
        if track:
            if nq_local > 0:
                trk = net.task_heads["track_2d"]
                if hasattr(trk, "join_streams"):
                    trk.defer_join = trk.own_stream = os.environ.get("L4P_TRACK_DEFER", "1") != "0"
                    trk.start_event = ready if trk.own_stream else None
                    ts = cu_masked_stream(net.device, os.environ.get("L4P_C5_TRK_CUS")) if trk.own_stream else None
                    trk.clip_stream_override = [ts] if ts is not None else None
                wins_t = [DecodedWindow(net.cfg.depth, {}, g["last"]) for g in lasts]
                trk_out = trk.forward_windowed(enc_features_bpc_2dlist=wins_t, time_strides=strides, **d_trk)
        if dec_join is not None:  # (the tracker is queued: now the main stream may wait for the decoders)
            dec_join()
            dec_join = None
            del groups
        out: dict = {}
        seam_local = dense and os.environ.get("L4P_C5_EXCHANGE", "gather") == "seam" and seam_local_supported(net, dense)
        if seam_local:
            # SURVEY.md 8e's exchange: K broadcast + one tail per chunk boundary + 18 floats per seam instead of every decoded window
            own = {w: DecodedWindow(net.cfg.depth, {k[4:]: v for k, v in g.items() if k.startswith("dec.")}, None)
                   for w, g in local.items()}
            if trk is not None and hasattr(trk, "join_streams") and not stitch_beside_tracker():
                trk.join_streams()
            out = stitch_seam_local(net, data, dense, own, rank, world)
        elif dense:
            gathered = all_gather_windows(local, nwin, rank, world)  # the exchange step of the dense path
            windows = [DecodedWindow(net.cfg.depth, {k[4:]: v for k, v in g.items() if k.startswith("dec.")}, None)
                       for g in gathered]
            if trk is not None and hasattr(trk, "join_streams") and not stitch_beside_tracker():
                trk.join_streams()
            out = net.stitch_windows(windows, data, dense, strides)
    finally:
        if dec_join is not None:  # (an exception before the join: the decoders' stream may still read the encoder's hook features)
            dec_join()
        if trk is not None and hasattr(trk, "join_streams"):
            trk.join_streams()
            trk.defer_join = trk.own_stream = False
            trk.start_event = None
            trk.clip_stream_override = None
    if trk_out is not None:
        out.update(trk_out)
    if track and _collectives_on(world):
        nq = data["track_2d_pointquerries_bn3"].shape[1]
        name = net.task_heads["track_2d"].task_name
        for key, shp in ((f"{name}_traj_est_bn2t", 2), (f"{name}_vis_est_bn1t", 1), (f"{name}_depth_est_bn1t", 1)):
            if key not in out:  # rank without queries
                out[key] = torch.zeros(data["rgb_b3thw"].shape[0], 0, shp, T, dtype=torch.float32, device=net.device)
            out[key] = all_gather_queries(out[key], nq, rank, world, dim=1)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Seam-local exchange of the dense path (option of forward_windows_sharded, L4P_C5_EXCHANGE=seam).
#
# The reference aligns every window to the ACCUMULATED buffer, one window after the other (dense_heads.py:444-467) - a chain
# over all windows that the default schedule above reproduces bit for bit by gathering every decoded window (12.9 MB each) on
# every rank and repeating the stitch there.  A similarity estimate is equivariant under a similarity of its target, though:
# aligning window w to the accumulated window w - 1 is aligning it to the RAW window w - 1 and composing with window w - 1's
# accumulated transform.  So:
#   1. window 0's owner estimates K (the camray head's fixed intrinsics)                 -> broadcast, 16 floats per clip
#   2. every rank runs the heads of its own windows (raw depth / poses, flow, mask)
#   3. the last 8 frames of a chunk's last window (depth + poses + K, one flow frame) go to the NEXT rank  -> P2P, 2.0 MB per clip
#   4. every rank solves its own seams against the raw predecessor (l4p_similarity_ransac)
#   5. all-gather of the seam records                                                     -> 18 floats per seam and clip
#   6. prefix composition (l4p_similarity_prefix), applied to the rank's own windows; the frames a window is the last to write
#      stay on the rank that produced them (outputs are gathered only if the caller asks for whole tensors)
# Not the reference's arithmetic to the bit: the RANSAC threshold is 0.01 x q98 of the raw predicted depth either way, but the
# residuals are measured in the target's units - raw here, accumulated there - so seams whose accumulated scale is far from 1
# draw a (slightly) different consensus set; on well-posed windows the two schedules agree to 1e-3
# (tests/test_seam_local_gpu.py), and the gather schedule stays the default until an N > 1 run decides.
# The phases are plain functions of one rank's state; `_SeamComm` moves the three messages through torch.distributed, the tests
# run the same phases for every emulated rank of one process and hand the messages over by hand.
# ---------------------------------------------------------------------------------------------------------------------
class SeamLocalState:
    """What one rank holds between the phases: its decoded windows, their head outputs, its seam records."""

    def __init__(self, rank: int, world: int, windows: dict, strides, ws: int):
        self.rank, self.world, self.windows, self.strides, self.ws = rank, world, windows, [int(s) for s in strides], ws
        self.s0, self.e0 = window_chunks(len(self.strides), world)[rank]
        self.cur: dict = {}    # window id -> {"depth","camray","camray_intrinsics_est","flow_2d_backward","dyn_mask"}
        self.rel: dict = {}    # seam id w (window w against window w - 1) -> [B, 18]


def seam_local_supported(net, tasks: List[str]) -> bool:
    """The joint depth + camera path with the shipped aligner; flow / mask ride along (pure copies)."""
    dense = [t for t in tasks if t != "track_2d"]
    return (net.joint_alignment and "depth" in dense and "camray" in dense and
            all(t in ("depth", "camray", "flow_2d_backward", "dyn_mask") for t in dense))


def seam_phase_k0(net, st: SeamLocalState, data: dict, img_info) -> Optional[torch.Tensor]:
    """Phase 1 on window 0's owner: the camray head on window 0 (estimates K when the head does) -> K [B,4,4,T] or None."""
    cam = net.task_heads["camray"]
    if st.s0 != 0 or st.e0 == 0 or cam.use_intrinsics or not cam.fixed_intrinsics:
        return None
    cam.forward(st.windows[0], img_info=img_info, intrinsics_b44t=data["intrinsics_b44t"][..., :st.ws], win_id=0)
    return cam.first_window_intrinsics_b44t.clone()


def seam_phase_heads(net, st: SeamLocalState, data: dict, tasks: List[str], img_info, k0: Optional[torch.Tensor]) -> Optional[dict]:
    """Phase 2: head outputs of this rank's windows, each in its own frame.  Returns the message for the NEXT rank (None on the
    last rank / a rank without windows): the tail of this chunk's last window."""
    cam = net.task_heads["camray"]
    K_in = data["intrinsics_b44t"]
    for w in range(st.s0, st.e0):
        t0 = st.strides[w]
        cur = {}
        for name in [t for t in tasks if t != "track_2d"]:
            head = net.task_heads[name]
            if name == "camray" and w > 0 and k0 is not None:
                cam.first_window_intrinsics_b44t = k0  # (a later window reports window 0's estimate, dense_heads.py:327-333)
            o = head.forward(st.windows[w], img_info=img_info, intrinsics_b44t=K_in[..., t0:t0 + st.ws], win_id=w)
            cur[name] = o[f"{head.task_name}_est_{head.task_suffix}"]
            if name == "camray":
                kkey = f"{head.task_name}_intrinsics_est_{head.task_suffix}"
                cur["camray_intrinsics_est"] = (o[kkey] if kkey in o else K_in[..., t0:t0 + st.ws].clone().reshape(-1, 16, st.ws))
        st.cur[w] = cur
    if st.e0 == st.s0 or st.e0 == len(st.strides):
        return None
    last, nxt = st.e0 - 1, st.e0
    ov = st.strides[last] + st.ws - st.strides[nxt]
    c = st.cur[last]
    msg = {"depth": c["depth"][:, :, st.ws - ov:].contiguous(), "camray": c["camray"][:, :, st.ws - ov:].contiguous(),
           "camray_intrinsics_est": c["camray_intrinsics_est"][:, :, st.ws - ov:].contiguous()}
    if "flow_2d_backward" in c:  # the first frame of a later window's flow is invalid: it keeps the predecessor's (dense_heads.py:139)
        msg["flow_frame"] = c["flow_2d_backward"][:, :, st.ws - ov:st.ws - ov + 1].contiguous()
    return msg


def seam_phase_solve(st: SeamLocalState, prev_tail: Optional[dict], img_info) -> dict:
    """Phase 4: the similarity of each of this rank's windows w > 0 against the RAW window w - 1 (its tail came from the previous
    rank for the chunk's first window).  -> {w: [B, 18]}"""
    from .models.aligner import KabaschUmeyama3DAligner

    st.prev_tail = prev_tail
    for w in range(max(st.s0, 1), st.e0):
        ov = st.strides[w - 1] + st.ws - st.strides[w]
        if w - 1 >= st.s0:
            p = st.cur[w - 1]
            tgt = {"depth": p["depth"][:, :, st.ws - ov:], "camray": p["camray"][:, :, st.ws - ov:],
                   "camray_intrinsics": p["camray_intrinsics_est"][:, :, st.ws - ov:].reshape(-1, 4, 4, ov)}
        else:
            assert prev_tail is not None, f"rank {st.rank}: the tail of window {w - 1} did not arrive"
            tgt = {"depth": prev_tail["depth"], "camray": prev_tail["camray"],
                   "camray_intrinsics": prev_tail["camray_intrinsics_est"].reshape(-1, 4, 4, ov)}
        c = st.cur[w]
        pred = {"depth": c["depth"][:, :, :ov], "camray": c["camray"][:, :, :ov],
                "camray_intrinsics": c["camray_intrinsics_est"][:, :, :ov].reshape(-1, 4, 4, ov).clone()}
        al = KabaschUmeyama3DAligner()
        al.solve(pred, tgt, img_info)
        st.rel[w] = al.rel_T_b44
    return st.rel


def seam_phase_apply(st: SeamLocalState, rel_all: torch.Tensor) -> dict:
    """Phases 6: ``rel_all`` [n_windows - 1, B, 18] (seam w - 1 = window w against raw window w - 1) -> prefix composition ->
    this rank's windows in window 0's frame -> the frames this rank is the last to write: {key: [B, C, frames, ...]} plus
    "frame_range": (f0, f1).  Window w is the last writer of frames [stride_w, stride_{w+1}) (the last window: to the end)."""
    from . import _lib
    from .ops import _p, _stream
    from .utils.umeyama import apply_window_similarity

    nwin = len(st.strides)
    if st.e0 == st.s0:
        return {"frame_range": (0, 0)}
    B = rel_all.shape[1]
    acc = torch.empty(nwin, B, 18, dtype=torch.float32, device=rel_all.device)
    _lib.check(_lib.load().l4p_similarity_prefix(_stream(), _p(rel_all.contiguous()), _p(acc), nwin - 1, B), "l4p_similarity_prefix")
    f0 = st.strides[st.s0]
    f1 = st.strides[st.e0] if st.e0 < nwin else st.strides[-1] + st.ws
    out = {}
    for w in range(st.s0, st.e0):
        c = st.cur[w]
        al = apply_window_similarity(acc[w], {"depth": c["depth"], "camray": c["camray"]}) if w > 0 else c
        a = st.strides[w] - f0
        n = (st.strides[w + 1] if w + 1 < nwin else st.strides[w] + st.ws) - st.strides[w]
        parts = {"depth": al["depth"], "camray": al["camray"], "camray_intrinsics_est": c["camray_intrinsics_est"]}
        for k in ("flow_2d_backward", "dyn_mask"):
            if k in c:
                parts[k] = c[k]
        for k, v in parts.items():
            if k not in out:
                shp = list(v.shape)
                shp[2] = f1 - f0
                out[k] = torch.zeros(*shp, dtype=v.dtype, device=v.device)
            out[k][:, :, a:a + n] = v[:, :, :n]
        if "flow_2d_backward" in c and w > 0:  # frame 0 of a later window's flow is the predecessor's frame at that time
            src = st.cur[w - 1]["flow_2d_backward"][:, :, st.strides[w] - st.strides[w - 1]] if w - 1 >= st.s0 else st.prev_tail["flow_frame"][:, :, 0]
            out["flow_2d_backward"][:, :, a] = src
    out["frame_range"] = (f0, f1)
    return out


def seam_exchange_bytes(B: int, nwin: int, world: int, ws: int = 16, H: int = 224, W: int = 224, ov: int = 8,
                        tasks=("depth", "camray", "flow_2d_backward", "dyn_mask")) -> dict:
    """Bytes one (interior) rank RECEIVES for the dense path under the two schedules (float32 tensors)."""
    per_win = 0
    for t in tasks:
        per_win += {"depth": ws * H * W, "dyn_mask": ws * H * W, "flow_2d_backward": 2 * ws * H * W, "camray": 6 * 16 * 16 * 16}[t] * 4
    own = window_chunks(nwin, world)[min(1, world - 1)]
    gather = B * per_win * (nwin - (own[1] - own[0]))
    tail = B * ((ov * H * W + 2 * 16 * ov) * 4 + (2 * H * W * 4 if "flow_2d_backward" in tasks else 0))
    seam = B * (16 * ws * 4 + tail + (nwin - 1) * 18 * 4)
    return {"gather_schedule": int(gather), "seam_local_schedule": int(seam)}


class _SeamComm:
    """The three messages of the seam-local schedule through torch.distributed (gloo in the CPU tests, RCCL on a node)."""

    def __init__(self, rank: int, world: int, device: torch.device):
        self.rank, self.world, self.device = rank, world, device

    def bcast_k0(self, k0: Optional[torch.Tensor], B: int, T: int) -> torch.Tensor:
        buf = k0.contiguous() if k0 is not None else torch.empty(B, 4, 4, T, dtype=torch.float32, device=self.device)
        dist.broadcast(buf, src=0)
        return buf

    def pass_tail(self, msg: Optional[dict], like: Optional[dict], has_prev: bool, has_next: bool) -> Optional[dict]:
        """Send ``msg`` to rank + 1 (if any), receive the previous rank's into tensors shaped ``like``."""
        ops, got = [], None
        if has_next and msg is not None:
            for k in sorted(msg):
                ops.append(dist.P2POp(dist.isend, msg[k], self.rank + 1))
        if has_prev:
            got = {k: torch.empty_like(v) for k, v in like.items()}
            for k in sorted(got):
                ops.append(dist.P2POp(dist.irecv, got[k], self.rank - 1))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        return got

    def gather_seams(self, rel: dict, nwin: int, B: int) -> torch.Tensor:
        chunks = window_chunks(nwin, self.world)
        cmax = max(e - s for s, e in chunks)
        s0, e0 = chunks[self.rank]
        block = torch.zeros(cmax, B, 18, dtype=torch.float32, device=self.device)
        for j, w in enumerate(range(s0, e0)):
            if w in rel:
                block[j].copy_(rel[w])
        parts = [torch.empty_like(block) for _ in range(self.world)]
        dist.all_gather(parts, block)
        rows = [parts[r][j] for r, (s, e) in enumerate(chunks) for j, w in enumerate(range(s, e)) if w > 0]
        return torch.stack(rows, dim=0) if rows else torch.zeros(0, B, 18, dtype=torch.float32, device=self.device)


def stitch_seam_local(net, data: dict, tasks: List[str], windows: dict, rank: int, world: int, gather_outputs: bool = True) -> dict:
    """The dense path of one rank under the seam-local schedule (``windows``: {window id: DecodedWindow} of this rank's chunk).
    ``gather_outputs``: all-gather the frame blocks into whole [B, C, T, ...] tensors on every rank (what the caller of
    L4P_VideoMAE.forward expects); False: every rank keeps the frames it produced ("frame_range")."""
    T = data["rgb_b3thw"].shape[2]
    B = data["rgb_b3thw"].shape[0]
    img_info = tuple(data.get("img_info", net.window_size))
    strides = net.time_strides(T)
    nwin = len(strides)
    st = SeamLocalState(rank, world, windows, strides, net.window_size[0])
    on = _collectives_on(world)
    comm = _SeamComm(rank, world, net.device) if on else None
    k0 = seam_phase_k0(net, st, data, img_info)
    cam = net.task_heads["camray"]
    if comm is not None and not cam.use_intrinsics and cam.fixed_intrinsics:
        k0 = comm.bcast_k0(k0, B, st.ws)
    msg = seam_phase_heads(net, st, data, tasks, img_info, k0)
    prev = None
    if comm is not None:
        chunks = window_chunks(nwin, world)
        has_prev = rank > 0 and st.e0 > st.s0 and st.s0 > 0
        has_next = msg is not None and rank + 1 < world and chunks[rank + 1][1] > chunks[rank + 1][0]
        like = None
        if has_prev:
            ov = st.strides[st.s0 - 1] + st.ws - st.strides[st.s0]
            c = st.cur[st.s0]
            like = {"depth": c["depth"][:, :, :ov], "camray": c["camray"][:, :, :ov],
                    "camray_intrinsics_est": c["camray_intrinsics_est"][:, :, :ov]}
            if "flow_2d_backward" in c:
                like["flow_frame"] = c["flow_2d_backward"][:, :, :1]
            like = {k: v.contiguous() for k, v in like.items()}
        prev = comm.pass_tail(msg, like, has_prev, has_next)
    rel = seam_phase_solve(st, prev, img_info)
    if comm is not None:
        rel_all = comm.gather_seams(rel, nwin, B)
    else:
        rel_all = (torch.stack([rel[w] for w in range(1, nwin)], dim=0) if nwin > 1 else
                   torch.zeros(0, B, 18, dtype=torch.float32, device=net.device))
    local = seam_phase_apply(st, rel_all)
    return seam_outputs(net, local, T, rank, world, nwin, gather_outputs and on)


def stitch_seam_local_emulated(net, data: dict, tasks: List[str], windows: list, world: int):
    """The phases of stitch_seam_local for EVERY rank of ``world`` in one process, the three messages handed over by hand (the
    emulated-rank tests and bench.py's one-of-eight-ranks measurement).  ``windows``: DecodedWindow of all windows.
    -> (frame blocks per rank, the seam records [n_windows - 1, B, 18])."""
    T = data["rgb_b3thw"].shape[2]
    B = data["rgb_b3thw"].shape[0]
    img_info = tuple(data.get("img_info", net.window_size))
    strides = net.time_strides(T)
    nwin = len(strides)
    states = []
    for r in range(world):
        s0, e0 = window_chunks(nwin, world)[r]
        states.append(SeamLocalState(r, world, {w: windows[w] for w in range(s0, e0)}, strides, net.window_size[0]))
    k0 = seam_phase_k0(net, states[0], data, img_info)                                  # broadcast
    msgs = [seam_phase_heads(net, st, data, tasks, img_info, k0) for st in states]
    rel: dict = {}
    last_msg = None
    for r, st in enumerate(states):                                                      # P2P: the previous rank's tail
        rel.update(seam_phase_solve(st, last_msg, img_info))
        if msgs[r] is not None:
            last_msg = msgs[r]
    rel_all = (torch.stack([rel[w] for w in range(1, nwin)], dim=0) if nwin > 1 else
               torch.zeros(0, B, 18, dtype=torch.float32, device=net.device))            # all-gather of 18 floats per seam
    return [seam_phase_apply(st, rel_all) for st in states], rel_all


def seam_outputs(net, local: dict, T: int, rank: int, world: int, nwin: int, gather: bool) -> dict:
    """Output keys of L4P_VideoMAE.forward from a rank's frame blocks (optionally all-gathered into whole tensors)."""
    names = {"depth": "depth", "camray": "camray", "flow_2d_backward": "flow_2d_backward", "dyn_mask": "dyn_mask"}
    out = {}
    f0, f1 = local["frame_range"]
    for k, v in local.items():
        if k == "frame_range":
            continue
        if gather:
            chunks = window_chunks(nwin, world)
            lens = []
            strides = [int(s) for s in net.time_strides(T)]
            for s, e in chunks:
                lens.append(0 if e == s else (strides[e] if e < nwin else T) - strides[s])
            fmax = max(lens)
            xm = v.movedim(2, 0).contiguous()
            block = torch.zeros((fmax,) + tuple(xm.shape[1:]), dtype=v.dtype, device=v.device)
            block[: xm.shape[0]].copy_(xm)
            parts = [torch.empty_like(block) for _ in range(world)]
            dist.all_gather(parts, block)
            v = torch.cat([parts[r][:n] for r, n in enumerate(lens)], dim=0).movedim(0, 2).contiguous()
        if k == "camray_intrinsics_est":
            cam = net.task_heads["camray"]
            out[f"{cam.task_name}_intrinsics_est_{cam.task_suffix}"] = v
        else:
            head = net.task_heads[names[k]]
            out[f"{head.task_name}_est_{head.task_suffix}"] = v
    if not gather:
        out["frame_range"] = (f0, f1)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Collective self-test: callable on any lease with >= 2 ranks (backend "nccl" = RCCL over xGMI on a GPU node, "gloo" in
# the CPU tests).  bench.py runs it before the timed region when WORLD_SIZE > 1 and prints its verdict into the JSON line
# ("rccl_ranks", "rccl_selftest"), so a multi-GPU bench line is also evidence that the two collectives of the path ran.
# ---------------------------------------------------------------------------------------------------------------------
def collective_selftest(device: torch.device) -> dict:
    """Exercises exactly the collectives the path uses: (1) broadcast_weights of a packed arena with a known checksum,
    (2) all_gather_windows with UNEQUAL chunks (world + 1 windows: rank 0 owns two, every other rank one), (3) a MAX
    all-reduce as bench.py's timing uses.  Returns {"ranks", "backend", "ok", ...}; raises on a mismatch."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"ranks": 1, "backend": None, "ok": True}
    rank, world = dist.get_rank(), dist.get_world_size()
    backend = dist.get_backend()
    # (1) weight broadcast: a 3-tensor arena built on rank 0 only
    pw = None
    if rank == 0:
        from .packing import Packer

        pk = Packer(torch.bfloat16)
        g = torch.Generator().manual_seed(99)
        pk.T("a.w", torch.randn(130, 64, generator=g))
        pk.F("a.b", torch.randn(130, generator=g))
        pk.T("c.w", torch.randn(256, 32, generator=g), pad_rows=False)
        pw = PackedWeights.from_packer(pk, device, {"patch_kp": 1216})
    pw = broadcast_weights(pw, device)
    chk = int(pw.arena.to(torch.int64).sum().item())
    chks = [None] * world
    dist.all_gather_object(chks, (chk, int(pw.arena.numel()), sorted(pw.t.keys())))
    if any(c != chks[0] for c in chks):
        raise RuntimeError(f"weight broadcast mismatch across ranks: {chks}")
    # (2) unequal window chunks
    nwin = world + 1
    s0, e0 = window_chunks(nwin, world)[rank]
    local = {w: {"x": torch.full((2, 3), float(w), device=device), "y": torch.arange(4, device=device, dtype=torch.float32) + 10 * w}
             for w in range(s0, e0)}
    allw = all_gather_windows(local, nwin, rank, world)
    for w in range(nwin):
        if float(allw[w]["x"].mean()) != float(w) or float(allw[w]["y"][0]) != 10.0 * w:
            raise RuntimeError(f"all_gather_windows returned the wrong block for window {w} on rank {rank}")
    # (3) MAX all-reduce
    t = torch.tensor([float(rank)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t.item()) != world - 1:
        raise RuntimeError("all_reduce(MAX) wrong")
    return {"ranks": world, "backend": backend, "ok": True, "arena_checksum": chk, "windows_gathered": nwin}
