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
