"""Work that runs on several HIP streams at once must be reproducible (round-5 tests).

Round 4 saw the seam alignment compute zeros when it ran beside the tracker's streams.  The cause (profiles/
r05_stream_race_root_cause.md) is a gfx950 interaction: while a wave issues MFMAs, a packed-FP32 instruction with op_sel[1] = 1 on
another wave of the same SIMD reads src1 as zero in lanes 48..63.  The library no longer contains that instruction form
(tools/check_isa.py gates the link; tests/test_host_cpu.py runs the lint).  These tests hold the schedules that exposed it:

 * the alignment's pointmap kernel beside a stream of MFMA GEMMs (the kernel round 4 saw fail, the strongest trigger);
 * B = 2 clips x 3 windows x all five tasks with the joint depth / camera alignment - the tracker's clip streams run beside the
   dense decoders AND the seam alignment - fifty forwards, bit-identical, and equal to the two clips' own B = 1 forwards;
 * the sharded long-video path with B = 2 and two emulated ranks: the query shard is a COPY there, which the tracker's streams
   must not read before it exists (round-4 advisor finding), with the seam alignment beside the recursion.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd import _lib, ops, parallel
from l4p_amd.ops import _p, _stream
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build

TASKS = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]
TRACK = ["track_2d_traj_est_bn2t", "track_2d_vis_est_bn1t", "track_2d_depth_est_bn1t"]


@pytest.fixture(scope="module")
def mini():
    cfg = ModelCfg.mini()
    return cfg, seeded_state_dict(cfg)


def two_clip_batch(T: int, nq: int) -> dict:
    """Two DIFFERENT clips with different queries (clip 1: the mirrored video, the queries shifted)."""
    b = make_batch(T, nq)
    q2 = b["track_2d_pointquerries_bn3"].clone()
    q2[..., 1:] = 224.0 - q2[..., 1:]
    return {"rgb_b3thw": torch.cat([b["rgb_b3thw"], b["rgb_b3thw"].flip(-1)], dim=0),
            "intrinsics_b44t": b["intrinsics_b44t"].repeat(2, 1, 1, 1),
            "track_2d_pointquerries_bn3": torch.cat([b["track_2d_pointquerries_bn3"], q2], dim=0),
            "track_2d_pointlabels_bn": b["track_2d_pointlabels_bn"].repeat(2, 1)}


def test_pointmap_kernel_beside_a_stream_of_mfma_gemms(dev):
    """l4p_point_map_samples on the main stream while bf16 GEMMs (v_mfma_f32_16x16x32_bf16) fill the chip from a second stream:
    400 launches, every output equal to a launch made alone.  (Round-4 library: 5 - 90 % of the launches differ.)"""
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    F, H, W, ratio, seed = 3, 224, 224, 10, 20250213
    n = F * ((H * W) // ratio)
    depth = torch.rand(F, H, W, generator=g).add_(0.5).cuda()
    K = torch.eye(4).repeat(F, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 200.0
    K[:, 0, 2] = K[:, 1, 2] = 112.0
    K = K.reshape(F, 16).cuda()
    R = torch.linalg.qr(torch.randn(F, 3, 3, generator=g)).Q
    P = torch.eye(4).repeat(F, 1, 1)
    P[:, :3, :3] = R
    P[:, :3, 3] = torch.randn(F, 3, generator=g)
    P = P.reshape(F, 16).cuda()

    def pointmap():
        a = torch.empty(n, 3, device="cuda")
        _lib.check(lib.l4p_point_map_samples(_stream(), _p(depth), _p(K), _p(P), _p(a), F, H, W, ratio, seed), "l4p_point_map_samples")
        return a

    ref = pointmap().clone()
    x = (torch.randn(8192, 1408, generator=g) * 0.5).cuda().to(torch.bfloat16)
    w = ops.pad_rows((torch.randn(1408, 1408, generator=g) * 0.03).cuda().to(torch.bfloat16), 256)
    ops.gemm(x, w, 1408)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    bad = 0
    for rep in range(5):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            keep = [ops.gemm(x, w, 1408) for _ in range(60)]
        res = [pointmap() for _ in range(80)]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(r, ref)) for r in res)
        del keep, res
    assert bad == 0, f"{bad} of 400 launches differ from the launch made alone"


def test_two_clips_three_windows_all_tasks_fifty_forwards(dev, mini):
    """B = 2, T = 32 (3 windows), all five tasks, joint alignment on (configs/model.yaml): the per-clip tracker streams run beside
    the dense decoders and beside joint_windowed_estimation's seam alignment.  50 forwards bit-identical (bf16 engine); every clip equal to its
    own B = 1 forward (f32 engine, 1e-3 relative to the maximum; integer-valued fills exactly)."""
    cfg, sd = mini
    model = build(cfg, sd, "bf16")
    assert model.l4p_model.joint_alignment
    batch = two_clip_batch(32, 6)
    with torch.no_grad():
        first = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
        first = {k: v.clone() for k, v in first.items() if torch.is_tensor(v)}
        for it in range(49):
            again = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
            torch.cuda.synchronize()
            for k, v in first.items():
                assert torch.equal(again[k], v), (it, k)
    # ... and equal to the clips' own B = 1 forwards: in the exact-f32 engine, where a different batch size (GEMM tile / split-K
    # choice) only moves float summation order
    del model
    model = build(cfg, sd, "32-true")
    with torch.no_grad():
        both = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
        singles = [model.forward({k: v[b:b + 1].clone() for k, v in batch.items()}, TASKS) for b in range(2)]
    torch.cuda.synchronize()
    for k, v in both.items():
        if not torch.is_tensor(v):
            continue
        for b in range(2):
            one, got = singles[b][k][0], v[b]
            if k in TRACK:
                assert torch.equal(got == -10.0, one == -10.0) and torch.equal(got == 0.0, one == 0.0), (k, b)
            err = float((got - one).abs().max() / (one.abs().max() + 1e-30))
            assert err <= 1e-3, (k, b, err)


@pytest.mark.parametrize("beside", ["0", "1"])
def test_sharded_two_clips_two_emulated_ranks(dev, mini, monkeypatch, beside):
    """forward_windows_sharded with B = 2 on two emulated ranks (the all-gathers replaced by a merge with the other rank's
    windows, computed beforehand): the query shard [:, q0:q1] is a copy for B > 1 and must exist before the tracker's streams
    start.  Dense outputs bit-identical to the single-GPU forward on both ranks, the tracks' shards put together equal to it to
    rounding; with the seam alignment behind (0) and beside (1) the tracker recursion; three times over."""
    cfg, sd = mini
    model = build(cfg, sd, "bf16")
    net = model.l4p_model
    batch = two_clip_batch(32, 5)
    monkeypatch.setenv("L4P_TRACK_BESIDE_STITCH", beside)
    world, nwin = 2, 3
    with torch.no_grad():
        ref = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
        data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        everything = {}
        for r in range(world):
            everything.update(parallel.decode_local_windows(net, data, TASKS, r, world))
        torch.cuda.synchronize()

        def emulated_gather(local, n_windows, rank, world_):
            keys = set(next(iter(local.values())).keys()) if local else None
            return [local[w] if w in local else {k: v for k, v in everything[w].items() if keys is None or k in keys}
                    for w in range(n_windows)]

        monkeypatch.setattr(parallel, "all_gather_windows", emulated_gather)
        for rep in range(3):
            outs = [parallel.forward_windows_sharded(net, {k: v.clone() for k, v in batch.items()}, TASKS, rank=r, world=world)
                    for r in range(world)]
            torch.cuda.synchronize()
            for key, val in ref.items():
                if not torch.is_tensor(val):
                    continue
                if key in TRACK:
                    got = torch.cat([o[key] for o in outs], dim=1)
                    assert got.shape == val.shape, key
                    assert torch.equal(got == -10.0, val == -10.0), (key, rep)
                    rl2 = float((got - val).norm() / val.norm())
                    assert rl2 <= 3e-2, (key, rep, rl2)
                else:
                    for r in range(world):
                        assert torch.equal(outs[r][key], val), (key, r, rep)


@pytest.mark.parametrize("T", [16, 32])
def test_dense_decoders_on_their_own_streams_equal_serial(dev, mini, monkeypatch, T):
    """The motion-mask and flow decoders run on streams of their own beside the depth / camray decoders and the tracker's clip
    streams (L4P_VideoMAE._run_heads_on_streams; L4P_HEAD_STREAMS=0 = one after the other): every output bit-identical to the
    serial order, over one window and over three (their seam alignment runs on the side streams too), B = 2, twenty forwards -
    no buffer goes back to the allocator under a stream that still reads it."""
    cfg, sd = mini
    model = build(cfg, sd, "bf16")
    batch = two_clip_batch(T, 5)
    with torch.no_grad():
        monkeypatch.setenv("L4P_HEAD_STREAMS", "0")
        serial = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
        serial = {k: v.clone() for k, v in serial.items() if torch.is_tensor(v)}
        monkeypatch.setenv("L4P_HEAD_STREAMS", "1")
        for it in range(20):
            out = model.forward({k: v.clone() for k, v in batch.items()}, TASKS)
            junk = [torch.randn(1 << 20, device="cuda") for _ in range(4)]  # (allocator pressure on the main stream's pool)
            torch.cuda.synchronize()
            for k, v in serial.items():
                assert torch.equal(out[k], v), (it, k)
            del out, junk
    assert getattr(model.l4p_model, "_head_streams", None), "the side streams were never used"
