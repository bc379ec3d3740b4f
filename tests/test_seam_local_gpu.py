"""The seam-local exchange of the sharded long video (l4p_amd/parallel.py) against the sequential stitch.

The reference aligns every window to the ACCUMULATED buffer, one after the other (dense_heads.py:444-467); the default sharded
schedule reproduces that bit for bit by gathering every decoded window on every rank.  The seam-local schedule solves each seam
against the RAW neighbour on the rank that owns the window, all-gathers 18 floats per seam and composes prefixes
(l4p_similarity_prefix); every rank then transforms its own windows and keeps the frames it produced.

  * l4p_similarity_prefix == the 4x4 products in float64, and applying a composed record == applying the two records in a row;
  * on WELL-POSED windows (tests/test_wellposed_heads_gpu._scene: a rendered camera path, 4 windows in their own frames and
    scales 1.0 / 1.7 / 0.6 / 1.25, measurement noise, gross depth outliers) the schedule on 2 and on 4 emulated ranks equals the
    sequential stitch to 1e-3 for depth / poses / K - the similarity estimate is equivariant, the consensus sets differ only by
    points next to the threshold - and BIT FOR BIT for flow and motion mask (pure copies, incl. the flow's frame-0 rule);
  (the messages themselves through torch.distributed and the byte counts of the two schedules: tests/test_parallel_cpu.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd import _lib, parallel
from l4p_amd.ops import _p, _stream
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.test_wellposed_heads_gpu import H, T, W, WS, _scene


def _records(n, B, seed):
    g = torch.Generator().manual_seed(seed)
    rec = torch.zeros(n, B, 18)
    for i in range(n):
        for b in range(B):
            R = torch.linalg.qr(torch.randn(3, 3, generator=g)).Q
            if torch.linalg.det(R) < 0:
                R[:, 0] = -R[:, 0]
            s = float(0.6 + torch.rand(1, generator=g))
            M = torch.eye(4)
            M[:3, :3] = s * R
            M[:3, 3] = torch.randn(3, generator=g)
            rec[i, b, :16] = M.reshape(16)
            rec[i, b, 16] = s
            rec[i, b, 17] = 1000 + i
    return rec


def test_similarity_prefix_is_the_product_of_the_records(dev):
    lib = _lib.load()
    n, B, Tt = 7, 3, 5
    rel = _records(n, B, 3).cuda()
    acc = torch.empty(n + 1, B, 18, device="cuda")
    _lib.check(lib.l4p_similarity_prefix(_stream(), _p(rel), _p(acc), n, B), "l4p_similarity_prefix")
    acc = acc.cpu()
    for b in range(B):
        A, s = np.eye(4), 1.0
        assert torch.equal(acc[0, b, :16].reshape(4, 4), torch.eye(4)) and float(acc[0, b, 16]) == 1.0
        for w in range(n):
            A = A @ rel[w, b, :16].reshape(4, 4).double().cpu().numpy()
            s *= float(rel[w, b, 16])
            assert np.abs(acc[w + 1, b, :16].reshape(4, 4).numpy() - A).max() <= 1e-5 * np.abs(A).max()
            assert abs(float(acc[w + 1, b, 16]) - s) <= 1e-6 * s
    # applying the composed record == applying rel[1] first and rel[0] after it (window 2 -> window 1's frame -> window 0's)
    g = torch.Generator().manual_seed(9)
    pose = torch.randn(16, Tt, generator=g)
    for t in range(Tt):
        R = torch.linalg.qr(torch.randn(3, 3, generator=g)).Q
        P = torch.eye(4)
        P[:3, :3], P[:3, 3] = R, torch.randn(3, generator=g)
        pose[:, t] = P.reshape(16)
    depth = torch.rand(1000, generator=g) + 0.5
    p1, d1 = pose.clone().cuda(), depth.clone().cuda()
    for w in (1, 0):
        _lib.check(lib.l4p_similarity_apply(_stream(), rel[w, 0].data_ptr(), p1.data_ptr(), Tt, d1.data_ptr(), d1.numel()), "apply")
    p2, d2 = pose.clone().cuda(), depth.clone().cuda()
    a2 = acc[2, 0].cuda()
    _lib.check(lib.l4p_similarity_apply(_stream(), a2.data_ptr(), p2.data_ptr(), Tt, d2.data_ptr(), d2.numel()), "apply")
    torch.cuda.synchronize()
    assert (p1 - p2).abs().max() <= 1e-5 * p1.abs().max() and (d1 - d2).abs().max() <= 1e-6 * d1.abs().max()


def _emulate(net, data, tasks, windows, world):
    """parallel.stitch_seam_local_emulated + the frame blocks of the ranks put back together."""
    blocks, rel_all = parallel.stitch_seam_local_emulated(net, data, tasks, windows, world)
    ranges = [b["frame_range"] for b in blocks]
    assert ranges[0][0] == 0 and ranges[-1][1] == T and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])), ranges
    out = {k: torch.cat([b[k] for b in blocks], dim=2) for k in blocks[0] if k != "frame_range"}
    return out, rel_all


@pytest.mark.parametrize("world", [2, 4])
def test_seam_local_schedule_equals_the_sequential_stitch_on_rendered_windows(dev, world):
    from tests.test_encoder_dpt_gpu import build

    cfg = ModelCfg.mini()
    model = build(cfg, seeded_state_dict(cfg), "32-true")
    net = model.l4p_model
    cam = net.task_heads["camray"]
    cam.use_intrinsics = False  # as shipped: K from window 0's ray map, later windows report it
    K, c2w, depth, wins = _scene()
    strides = net.time_strides(T)
    g = torch.Generator().manual_seed(21)
    tasks = ["depth", "camray", "flow_2d_backward", "dyn_mask"]
    assert parallel.seam_local_supported(net, tasks)
    windows = [parallel.DecodedWindow(cfg.depth, {"depth": w["depth"].cuda(), "camray": w["rays"].cuda(),
                                                  "flow_2d_backward": torch.randn(1, 2, WS, H, W, generator=g).cuda(),
                                                  "dyn_mask": torch.randn(1, 1, WS, H, W, generator=g).cuda()}, None) for w in wins]
    data = {"rgb_b3thw": torch.zeros(1, 3, T, H, W, device="cuda"),
            "intrinsics_b44t": K[None, :, :, None].repeat(1, 1, 1, T).cuda()}
    with torch.no_grad():
        seq = net.stitch_windows(windows, data, tasks, strides)
        loc, rel_all = _emulate(net, data, tasks, windows, world)
    torch.cuda.synchronize()
    pairs = {"depth": "depth_est_b1thw", "camray": "traj3d_est_b16t", "camray_intrinsics_est": "traj3d_intrinsics_est_b16t"}
    for k, key in pairs.items():
        y, r = loc[k].float().cpu(), seq[key].float().cpu()
        assert y.shape == r.shape, (k, y.shape, r.shape)
        e = float((y - r).abs().max() / r.abs().max())
        print(f"world {world}: {key} seam-local vs sequential {e:.2e}")
        assert e <= 1e-3, (k, e)
    for k, key in (("flow_2d_backward", "flow_2d_backward_est_b2thw"), ("dyn_mask", "dyn_mask_est_b1thw")):
        assert torch.equal(loc[k], seq[key]), k
    # the seam records are RELATIVE ones: scales close to the ratios of the rendered windows' scales (1.0, 1.7, 0.6, 1.25)
    s = rel_all[:, 0, 16].cpu().numpy()
    want = np.array([1.0 / 1.7, 1.7 / 0.6, 0.6 / 1.25])
    assert np.abs(s / want - 1).max() <= 2e-2, (s, want)
