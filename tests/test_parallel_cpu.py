"""CPU, world_size 2, gloo: the only collective on the path (weight-arena broadcast) and the clip sharding."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from l4p_amd import packing
    from l4p_amd.parallel import broadcast_weights, init_distributed, shard
    from l4p_amd.weights import ModelCfg, seeded_state_dict

    init_distributed("gloo")
    cfg = ModelCfg(dim=176, depth=1, heads=2, mlp_hidden=768, hooks=(1, 1, 1, 1))
    pw = None
    if rank == 0:
        pw = packing.pack_state_dict(seeded_state_dict(cfg, tasks=[]), cfg, torch.bfloat16, torch.device("cpu"), tasks=[])
    pw = broadcast_weights(pw, torch.device("cpu"))
    digest = float(pw["enc.blk0.fc1.w"].float().sum()) + float(pw["enc.pos"].sum())
    q.put((rank, digest, int(pw.arena.numel()), pw.meta["patch_kp"], shard(7, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, d0, n0, kp0, s0), (r1, d1, n1, kp1, s1) = res
    assert d0 == d1 and n0 == n1 and kp0 == kp1 == 1216
    assert sorted(s0 + s1) == list(range(7)) and not set(s0) & set(s1)


def _worker_windows(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from l4p_amd.parallel import all_gather_queries, all_gather_windows, init_distributed, shard_queries, window_chunks

    init_distributed("gloo")
    nwin, nq = 7, 5  # unequal chunks on both axes: windows 4 + 3, queries 3 + 2
    s, e = window_chunks(nwin, world)[rank]
    local = {w: {"dec.depth": torch.full((1, 1, 4, 3, 3), float(w)), "last": torch.arange(6.0).reshape(1, 2, 3) + 100 * w}
             for w in range(s, e)}
    got = all_gather_windows(local, nwin, rank, world)
    ok = len(got) == nwin and all(float(got[w]["dec.depth"].mean()) == float(w) and
                                  torch.equal(got[w]["last"], torch.arange(6.0).reshape(1, 2, 3) + 100 * w) for w in range(nwin))
    q0, q1 = shard_queries(nq, rank, world)
    full = torch.arange(nq * 2 * 4, dtype=torch.float32).reshape(1, nq, 2, 4)
    back = all_gather_queries(full[:, q0:q1].contiguous(), nq, rank, world, dim=1)
    q.put((rank, bool(ok), bool(torch.equal(back, full)), (s, e), (q0, q1)))
    dist.barrier()
    dist.destroy_process_group()


def test_window_and_query_exchange_world2():
    """The one exchange step of the window-sharded long-video path (config 5): unequal chunks, padded all-gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_windows, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True] and [r[2] for r in res] == [True, True]
    assert [r[3] for r in res] == [(0, 4), (4, 7)] and [r[4] for r in res] == [(0, 3), (3, 5)]


def test_window_chunks_cover():
    from l4p_amd.parallel import window_chunks

    assert [e - s for s, e in window_chunks(31, 8)] == [4, 4, 4, 4, 4, 4, 4, 3]
    for n, w in ((31, 8), (3, 8), (16, 4), (1, 2)):
        ch = window_chunks(n, w)
        assert ch[0][0] == 0 and ch[-1][1] == n and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))


def _worker_selftest(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from l4p_amd.parallel import collective_selftest, init_distributed

    init_distributed("gloo")
    q.put((rank, collective_selftest(torch.device("cpu"))))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_selftest_world3():
    """parallel.collective_selftest — what bench.py runs on the live backend before an N > 1 measurement (and reports as
    rccl_ranks / rccl_selftest) — on gloo with 3 ranks: weight broadcast checksum, all-gather of 4 windows in chunks of
    2 + 1 + 1, MAX all-reduce."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_selftest, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(3)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1]["ok"] and r[1]["ranks"] == 3 and r[1]["backend"] == "gloo" and r[1]["windows_gathered"] == 4 for r in res)
    assert len({r[1]["arena_checksum"] for r in res}) == 1


def _worker_solo_inside_group(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from l4p_amd import parallel
    from l4p_amd.parallel import all_gather_queries, all_gather_windows, init_distributed

    init_distributed("gloo")
    # L4P_VideoMAE.forward(window_batch > 1) calls the sharded path with (rank 0, world 1) from EVERY rank of a data-parallel
    # job (l4p_videomae.py: forward_windows_sharded(self, data, tasks, 0, 1)): it owns all windows and must not enter a collective
    nwin = 3
    local = {w: {"dec.depth": torch.full((1, 1, 2, 2, 2), float(10 * rank + w))} for w in range(nwin)}
    got = all_gather_windows(local, nwin, 0, 1)
    ok_w = len(got) == nwin and all(got[w]["dec.depth"] is local[w]["dec.depth"] for w in range(nwin))
    x = torch.arange(8.0).reshape(1, 4, 2) + rank
    ok_q = all_gather_queries(x, 4, 0, 1, dim=1) is x
    ok_flag = (not parallel._collectives_on(1)) and parallel._collectives_on(world) and parallel._collectives_on()
    try:
        parallel._collectives_on(world + 1)
        ok_err = False
    except ValueError:
        ok_err = True
    q.put((rank, bool(ok_w), bool(ok_q), bool(ok_flag), ok_err))
    dist.barrier()
    dist.destroy_process_group()


def test_single_rank_caller_inside_a_larger_group_stays_local():
    """Round-3 advisor finding: with window_batch > 1 every rank of a 2-rank job runs the window-sharded path on its own
    (world = 1); the exchange helpers then return the local tensors and never touch the group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_solo_inside_group, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1:] == (True, True, True, True) for r in res), res


def _worker_seam(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from l4p_amd.parallel import _SeamComm, init_distributed, window_chunks

    init_distributed("gloo")
    nwin, B, T = 7, 2, 16  # chunks 3 + 2 + 2
    comm = _SeamComm(rank, world, torch.device("cpu"))
    k0 = comm.bcast_k0(torch.arange(B * 16 * T, dtype=torch.float32).reshape(B, 4, 4, T) if rank == 0 else None, B, T)
    ok_k = bool(torch.equal(k0, torch.arange(B * 16 * T, dtype=torch.float32).reshape(B, 4, 4, T)))
    s0, e0 = window_chunks(nwin, world)[rank]
    msg = {"depth": torch.full((B, 1, 8, 3, 3), float(rank)), "camray": torch.full((B, 16, 8), 10.0 + rank),
           "camray_intrinsics_est": torch.full((B, 16, 8), 20.0 + rank), "flow_frame": torch.full((B, 2, 1, 3, 3), 30.0 + rank)}
    like = {k: torch.empty_like(v) for k, v in msg.items()}
    got = comm.pass_tail(msg if rank + 1 < world else None, like if rank > 0 else None, rank > 0, rank + 1 < world)
    ok_t = (got is None) if rank == 0 else all(float(got[k].mean()) == base + rank - 1 for k, base in
                                                (("depth", 0.0), ("camray", 10.0), ("camray_intrinsics_est", 20.0), ("flow_frame", 30.0)))
    rel = {w: torch.full((B, 18), float(w)) for w in range(max(s0, 1), e0)}
    allr = comm.gather_seams(rel, nwin, B)
    ok_r = tuple(allr.shape) == (nwin - 1, B, 18) and all(float(allr[w - 1].mean()) == float(w) for w in range(1, nwin))
    q.put((rank, ok_k, bool(ok_t), bool(ok_r)))
    dist.barrier()
    dist.destroy_process_group()


def test_seam_local_messages_world3():
    """The three messages of the seam-local exchange (parallel._SeamComm: K broadcast, one tail to the next rank, all-gather of the
    seam records over unequal chunks) through gloo with three ranks; and what a rank receives under the two schedules."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_seam, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(3))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] and r[3] for r in res), res
    from l4p_amd.parallel import seam_exchange_bytes

    b = seam_exchange_bytes(B=1, nwin=31, world=8)
    # every decoded window a rank does not own (27 x 12.9 MB) against one tail + K + 30 seam records
    assert 300e6 < b["gather_schedule"] < 400e6 and 1.5e6 < b["seam_local_schedule"] < 2.5e6, b


def test_decoder_stream_choice(monkeypatch):
    """parallel.decoder_stream: where the decoders of a rank's windows are queued while its query shard is tracked beside them -
    160 CUs for a small shard (<= 16 queries), the whole chip otherwise; L4P_C5_DEC_CUS overrides ("" / "0": the whole chip)."""
    import torch

    from l4p_amd import parallel as par

    seen = []
    monkeypatch.setattr(par, "cu_masked_stream", lambda dev, spec: seen.append(spec) or ("stream", spec))
    dev = torch.device("cpu")
    monkeypatch.delenv("L4P_C5_DEC_CUS", raising=False)
    assert par.decoder_stream(dev, 8) == ("stream", "0,160")
    assert par.decoder_stream(dev, 16) == ("stream", "0,160")
    assert par.decoder_stream(dev, 64) is None and par.decoder_stream(dev, 0) is None
    monkeypatch.setenv("L4P_C5_DEC_CUS", "")
    assert par.decoder_stream(dev, 8) is None
    monkeypatch.setenv("L4P_C5_DEC_CUS", "0")
    assert par.decoder_stream(dev, 8) is None
    monkeypatch.setenv("L4P_C5_DEC_CUS", "32,128")
    assert par.decoder_stream(dev, 64) == ("stream", "32,128")
    assert seen == ["0,160", "0,160", "32,128"]
