"""Config 5 (window-sharded long video): the two-phase sharded forward must reproduce the single-GPU windowed forward
bit for bit.  Two ranks are emulated on one GPU by running their phases one after the other and merging what the
collectives would exchange (the collectives themselves are covered on CPU/gloo in test_parallel_cpu.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd import parallel
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build

TASKS = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]
TRACK = ["track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t"]


@pytest.mark.parametrize("precision", ["32-true", "bf16"])
def test_sharded_windows_equal_single_gpu(dev, precision):
    cfg = ModelCfg.mini()
    model = build(cfg, seeded_state_dict(cfg), precision)
    net = model.l4p_model
    batch = make_batch(32, 5)  # 3 windows (stride 8), 5 queries: unequal shards on both axes
    with torch.no_grad():
        ref = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
        one = parallel.forward_windows_sharded(net, {k: v.clone() for k, v in batch.items()}, TASKS, rank=0, world=1)
        data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        world = 2
        local = [parallel.decode_local_windows(net, data, TASKS, r, world) for r in range(world)]
        assert sorted(local[0]) == [0, 1] and sorted(local[1]) == [2]
        gathered = [({**local[0], **local[1]})[w] for w in range(3)]
        outs = [parallel.stitch_gathered_windows(net, data, TASKS, gathered, r, world) for r in range(world)]
    torch.cuda.synchronize()
    for key, val in ref.items():
        if not torch.is_tensor(val):
            continue
        assert torch.equal(one[key], val), ("world 1", key)
        if key in TRACK:
            assert torch.equal(torch.cat([o[key] for o in outs], dim=1), val), ("world 2", key)
        else:
            for r in range(world):
                assert torch.equal(outs[r][key], val), ("world 2", r, key)


def test_grouped_windows_match_to_rounding(dev):
    """group = 4 (what bench.py --workload c5 runs): windows batched through the encoder / decoders; equal to the
    window-by-window forward up to summation order."""
    cfg = ModelCfg.mini()
    model = build(cfg, seeded_state_dict(cfg), "32-true")
    batch = make_batch(32, 5)
    with torch.no_grad():
        ref = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
        grp = parallel.forward_windows_sharded(model.l4p_model, {k: v.clone() for k, v in batch.items()}, TASKS, rank=0, world=1,
                                               group=4)
        model.l4p_model.window_batch = 4  # the same through the model's own forward (what demo/demo.py sets)
        own = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
        model.l4p_model.window_batch = 1
    torch.cuda.synchronize()
    for key, val in ref.items():
        if torch.is_tensor(val):
            assert (grp[key] - val).abs().max() <= 1e-4 * val.abs().max(), key
            assert torch.equal(own[key], grp[key]), key
