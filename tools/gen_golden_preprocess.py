"""Generate tests/golden/preprocess_clip.npz from the REAL reference data classes (runs only where /root/reference exists).

  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_preprocess.py

The reference's `l4p.data.video_dataset.VideoDataset` is imported from /root/reference and its own `__getitem__` is
run on seeded synthetic frames.  Three modules the image lacks are stubbed at import time:
  * mediapy        — only `VideoReader` is used (video_dataset.py:77): the stub yields the in-memory synthetic frames;
  * torchvision.transforms.functional — only `to_tensor` (video_dataset.py:93): uint8 HWC -> float32 CHW / 255;
  * kornia.morphology.erosion — imported by l4p_dataset_mini.py:13, never called for sampling version "uniform".
PIL (Pillow 12.2.0) and torch are the real ones.  Only data is written: the cases' parameters (inputs are regenerated
from seeds by tests/golden_utils.synthetic_video), SHA-256 digests of the blurred uint8 frames, sampled values and
statistics of the float outputs, and the small tensors (intrinsics, queries) in full.
"""
from __future__ import annotations

import hashlib
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np
import torch

from tests.golden_utils import PREPROCESS_CASES, sample_indices, synthetic_video

_VIDEOS = {}


class _Reader:
    def __init__(self, path):
        self.frames = _VIDEOS[path]
        self.num_images, self.shape, self.fps = len(self.frames), self.frames.shape[1:3], 30.0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __iter__(self):
        return iter(self.frames)


def install_stubs():
    media = types.ModuleType("mediapy")
    media.VideoReader = _Reader
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.to_tensor = lambda pic: torch.from_numpy(np.asarray(pic)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    tv.transforms, tvt.functional = tvt, tvf
    kornia = types.ModuleType("kornia")
    km = types.ModuleType("kornia.morphology")
    km.erosion = None
    kornia.morphology = km
    sys.modules.update({"mediapy": media, "torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf, "kornia": kornia, "kornia.morphology": km})
    sys.path.insert(0, REF)


def main():
    install_stubs()
    from PIL import Image

    from l4p.data.video_dataset import VideoDataset
    from oracle import preprocess_oracle as po

    out = {}
    for name, c in PREPROCESS_CASES.items():
        frames = synthetic_video(c["seed"], c["T"], c["H"], c["W"])
        _VIDEOS[name] = frames
        ds = VideoDataset(video_paths=[name], crop_size=c["crop_size"], estimation_directions=[1],
                          track_2d_querry_sampling_spacing=c["spacing"], max_frames=c["max_frames"], stride=c["stride"],
                          resize_size=tuple(c["resize_size"]))
        s = ds[0]
        rgb = s["rgb_b3thw"].numpy()
        # the blurred uint8 frames (video_dataset.py:86-92) straight from Pillow
        blurred = []
        for f in frames[: c["max_frames"] - 1]:
            im = Image.fromarray(f)
            full = im.size
            im = im.resize(tuple(c["resize_size"]), resample=Image.Resampling.BILINEAR)
            blurred.append(np.asarray(im.resize(full, resample=Image.Resampling.BILINEAR)))
        blurred = np.stack(blurred)
        idx = sample_indices(rgb.size, 4096).numpy()
        out[name + ".blur_sha256"] = np.frombuffer(hashlib.sha256(blurred.tobytes()).digest(), dtype=np.uint8)
        out[name + ".blur_frame0"] = blurred[0]
        out[name + ".rgb_shape"] = np.array(rgb.shape)
        out[name + ".rgb_idx"] = idx
        out[name + ".rgb_val"] = rgb.reshape(-1)[idx]
        out[name + ".rgb_stats"] = np.array([rgb.mean(dtype=np.float64), rgb.std(dtype=np.float64), np.abs(rgb).max()])
        out[name + ".intrinsics_b44t"] = s["intrinsics_b44t"].numpy()
        out[name + ".queries"] = s["track_2d_pointquerries_bn3"].numpy()
        out[name + ".labels"] = s["track_2d_pointlabels_bn"].numpy()
        out[name + ".ori_video_len"] = np.array(s["ori_video_len"])
        out[name + ".keys"] = np.array(sorted(k for k in s.keys()))
        # the restatement against the reference, full tensors
        o = po.preprocess_clip(frames, crop_size=c["crop_size"], resize_size=tuple(c["resize_size"]),
                               max_frames=c["max_frames"], stride=c["stride"], spacing=c["spacing"])
        err = float(np.abs(o["rgb_b3thw"] - rgb).max())
        assert o["rgb_b3thw"].shape == rgb.shape and err <= 2e-6, (name, err)
        assert np.array_equal(o["intrinsics_b44t"], out[name + ".intrinsics_b44t"]), name
        assert np.array_equal(o["track_2d_pointquerries_bn3"], out[name + ".queries"]), name
        assert int(o["ori_video_len"]) == int(s["ori_video_len"])
        ob = np.stack([po.resize_blur_resize(f, tuple(c["resize_size"])) for f in frames[: c["max_frames"] - 1]])
        assert np.array_equal(ob, blurred), name + ": PIL restatement is not bit-exact"
        out[name + ".oracle_max_abs_err"] = np.array(err)
        print(f"{name}: rgb {rgb.shape}  oracle max|err| {err:.2e}  blur bit-exact  queries {out[name + '.queries'].shape}")
    path = os.path.join(ROOT, "tests", "golden", "preprocess_clip.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
