"""Clip preparation on the GPU (csrc/preprocess.hip through the C ABI) against the oracle and the reference fixtures:
uint8 blur passes bit-exact with Pillow; the float tensor BIT-EQUAL to the oracle's un-contracted float32 restatement and
within 2e-6 abs of the reference's own output (ATen fuses some multiply-adds: 1-ulp lerp differences, amplified by 1/std;
values are O(1)); control values (intrinsics, queries, labels, lengths) exact."""
import hashlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd.data import VideoDataset, pil_resize_blur_resize, prepare_clip
from oracle import preprocess_oracle as po
from tests.golden_utils import PREPROCESS_CASES, synthetic_video

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "preprocess_clip.npz"))
TOL = 2e-6


@pytest.mark.parametrize("name", list(PREPROCESS_CASES))
def test_video_dataset_matches_oracle_and_reference_fixture(dev, name):
    c = PREPROCESS_CASES[name]
    frames = synthetic_video(c["seed"], c["T"], c["H"], c["W"])
    ds = VideoDataset(video_paths=["dir/" + name], crop_size=c["crop_size"], estimation_directions=[1],
                      track_2d_querry_sampling_spacing=c["spacing"], max_frames=c["max_frames"], stride=c["stride"],
                      resize_size=tuple(c["resize_size"]), frames={"dir/" + name: frames}, device=dev)
    s = ds[0]
    torch.cuda.synchronize()
    o = po.preprocess_clip(frames, crop_size=c["crop_size"], resize_size=tuple(c["resize_size"]), max_frames=c["max_frames"],
                           stride=c["stride"], spacing=c["spacing"])
    rgb = s["rgb_b3thw"].cpu().numpy()
    assert rgb.shape == o["rgb_b3thw"].shape and rgb.dtype == np.float32
    assert np.array_equal(rgb, o["rgb_b3thw"])                                              # full tensor, bit-equal to the oracle
    assert np.abs(rgb.reshape(-1)[GOLD[name + ".rgb_idx"]] - GOLD[name + ".rgb_val"]).max() <= TOL  # vs the reference itself
    assert np.array_equal(s["intrinsics_b44t"].cpu().numpy(), GOLD[name + ".intrinsics_b44t"])
    assert np.array_equal(s["track_2d_pointquerries_bn3"].cpu().numpy(), GOLD[name + ".queries"])
    assert np.array_equal(s["track_2d_pointlabels_bn"].cpu().numpy(), GOLD[name + ".labels"])
    assert s["ori_video_len"] == int(GOLD[name + ".ori_video_len"]) and s["seq_name"] == name
    assert sorted(s.keys()) == [str(k) for k in GOLD[name + ".keys"]]
    N, Tn = s["track_2d_pointquerries_bn3"].shape[0], rgb.shape[1]
    assert s["track_2d_traj_bn2t"].shape == (N, 2, Tn) and s["track_2d_vis_bn1t"].dtype == torch.bool
    assert s["instanceseg_b1thw"].shape == (1, Tn) + rgb.shape[2:] and float(s["track_2d_depth_bn1t"].min()) == 1.0


@pytest.mark.parametrize("name", list(PREPROCESS_CASES))
def test_blur_passes_bit_exact(dev, name):
    c = PREPROCESS_CASES[name]
    frames = synthetic_video(c["seed"], c["T"], c["H"], c["W"])[: c["max_frames"] - 1]
    out = pil_resize_blur_resize(torch.from_numpy(frames).to(dev), tuple(c["resize_size"])).cpu().numpy()
    assert np.array_equal(out[0], GOLD[name + ".blur_frame0"])
    assert hashlib.sha256(out.tobytes()).digest() == GOLD[name + ".blur_sha256"].tobytes()  # == Pillow on every byte


def test_fused_last_pass_equals_materialised_blur(dev):
    """The fused kernel (last vertical pass evaluated per output pixel) == blur written out, then resize: bitwise."""
    from l4p_amd import _lib
    from l4p_amd.data import video_dataset as vd
    import ctypes as C

    frames = torch.from_numpy(synthetic_video(3, 6, 270, 480)).to(dev)
    s = prepare_clip(frames, (8, 224, 224), (224, 224), spacing=0.5)
    blurred = pil_resize_blur_resize(frames, (224, 224))
    idx = torch.tensor(vd.mirror_pad_indices(6, 8)[:8], dtype=torch.int32, device=dev)
    rgb = torch.empty_like(s["rgb_b3thw"])
    mean, std = (C.c_float * 3)(*vd._MEAN), (C.c_float * 3)(*vd._STD)
    _lib.check(_lib.load().l4p_clip_resize_normalize(torch.cuda.current_stream().cuda_stream, blurred.data_ptr(), idx.data_ptr(),
                                                     rgb.data_ptr(), 8, 270, 480, 224, 224, 0, 0, 224, 224, mean, std, 270, None,
                                                     None, 0, None))
    torch.cuda.synchronize()
    assert torch.equal(rgb, s["rgb_b3thw"])


def test_full_size_video_properties_and_oracle_frames(dev):
    """DAVIS-sized input (480x854, 50 frames -> 64 after mirror padding): two frames against the oracle in full, and the
    size-independent properties of the path: a constant frame stays constant through the fixed-point blur, mirrored
    frames are bitwise copies of their sources, output is finite and normalised."""
    T, H, W = 50, 480, 854
    frames = synthetic_video(21, T, H, W)
    frames[7] = 93  # constant frame
    s = prepare_clip(torch.from_numpy(frames).to(dev), (64, 224, 224), (224, 224), spacing=0.04)
    torch.cuda.synchronize()
    rgb = s["rgb_b3thw"]
    assert rgb.shape == (3, 64, 224, 224) and bool(torch.isfinite(rgb).all())
    idx = po.mirror_pad_indices(T, 64)[:64]
    for t in (50, 57, 63):  # mirror-padded frames
        assert torch.equal(rgb[:, t], rgb[:, idx[t]])
    const = (np.float32(93) / np.float32(255) - po.IMAGENET_MEAN) / po.IMAGENET_STD
    assert np.abs(rgb[:, 7].cpu().numpy() - const[:, None, None]).max() <= TOL
    sub = frames[[0, 49]]
    o = po.preprocess_clip(sub, crop_size=(2, 224, 224), resize_size=(224, 224), spacing=0.5)
    assert np.array_equal(rgb[:, [0, 49]].cpu().numpy(), o["rgb_b3thw"])
    assert s["track_2d_pointquerries_bn3"].shape == (625, 3)


def test_error_behaviour(dev):
    with pytest.raises(ValueError):
        prepare_clip(torch.zeros((4, 8, 8, 3), device=dev), (16, 224, 224), (224, 224))  # not uint8
    with pytest.raises(AssertionError):
        prepare_clip(torch.zeros((4, 8, 8, 3), dtype=torch.uint8, device=dev), (16, 224, 224), (100, 100))  # crop > resized frame
    with pytest.raises(NotImplementedError):
        VideoDataset(video_paths=["x"], center_crop=False)
    with pytest.raises(ImportError):
        VideoDataset(video_paths=["missing.mp4"], device=dev)[0]  # no mediapy in this image and no decoded frames given
