"""RCCL on the hardware that is available: every box of the test pool has ONE MI355X, and RCCL refuses two ranks on one
device, so the N > 1 exchange of configs[3] / configs[4] cannot run here.  What can: a one-rank "nccl" (= RCCL) process group
with the path's collectives FORCED on (L4P_FORCE_COLLECTIVES=1, l4p_amd/parallel.py:_collectives_on), so that the very tensors
the multi-GPU path exchanges - the packed weight arena (uint8, one broadcast), the decoded windows and float last-layer
features (all-gather, bf16 / f32 blocks), the query-sharded tracks (all-gather), bench.py's MAX all-reduce - go through RCCL's
kernels on the GPU, with their real dtypes, devices and strides, and come back unchanged:
  * parallel.collective_selftest on backend "nccl";
  * broadcast_weights of the mini model's arena, then forward_windows_sharded over 3 windows == model.forward bit for bit.
Runs in a child process (the process group must not leak into the pytest process).  The world-2 / world-3 semantics of the same
functions are covered on gloo (tests/test_parallel_cpu.py) and by emulated ranks (tests/test_sharded_windows_gpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, os.environ["L4P_ROOT"])
import torch
import torch.distributed as dist
from l4p_amd import parallel
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build

rank, world, local = parallel.init_distributed("nccl")
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", local)
st = parallel.collective_selftest(dev)
assert st["ok"] and st["backend"] == "nccl" and st["ranks"] == 1 and st["windows_gathered"] == 2, st

TASKS = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]
cfg = ModelCfg.mini()
model = build(cfg, seeded_state_dict(cfg), "bf16")
net = model.l4p_model
before = int(net.weights.arena.to(torch.int64).sum().item())
w2 = parallel.broadcast_weights(net.weights, dev)   # the arena itself through RCCL's broadcast
assert int(w2.arena.to(torch.int64).sum().item()) == before
batch = make_batch(32, 5)
with torch.no_grad():
    ref = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
    out = parallel.forward_windows_sharded(net, {k: v.clone() for k, v in batch.items()}, TASKS)
torch.cuda.synchronize()
n = 0
for key, val in ref.items():
    if torch.is_tensor(val):
        assert torch.equal(out[key], val), key
        n += 1
t = torch.tensor([3.0], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.0
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"ok": True, "tensors_equal": n, "arena_bytes": int(w2.arena.numel()), "selftest": st}))
"""


def test_path_collectives_through_rccl_with_one_rank(dev):
    env = dict(os.environ)
    env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29617",
                "L4P_FORCE_COLLECTIVES": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "L4P_ROOT": ROOT})
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    assert '"ok": true' in r.stdout, r.stdout[-2000:]
