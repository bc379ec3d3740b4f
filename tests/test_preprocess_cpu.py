"""Clip preparation, CPU side: the oracle against the fixtures produced by the REAL reference
dataset classes + Pillow (tools/gen_golden_preprocess.py), against Pillow itself where importable, and the host
function of the C ABI (l4p_pil_coeffs) against the oracle's tables — bit-exact (integer work)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from oracle import preprocess_oracle as po
from tests.golden_utils import PREPROCESS_CASES, synthetic_video

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "preprocess_clip.npz"))


@pytest.mark.parametrize("name", list(PREPROCESS_CASES))
def test_oracle_matches_reference_fixture(name):
    c = PREPROCESS_CASES[name]
    frames = synthetic_video(c["seed"], c["T"], c["H"], c["W"])
    blurred = np.stack([po.resize_blur_resize(f, tuple(c["resize_size"])) for f in frames[: c["max_frames"] - 1]])
    assert np.array_equal(blurred[0], GOLD[name + ".blur_frame0"])
    assert hashlib.sha256(blurred.tobytes()).digest() == GOLD[name + ".blur_sha256"].tobytes()  # every frame, bit-exact
    o = po.preprocess_clip(frames, crop_size=c["crop_size"], resize_size=tuple(c["resize_size"]), max_frames=c["max_frames"],
                           stride=c["stride"], spacing=c["spacing"])
    rgb = o["rgb_b3thw"]
    assert tuple(rgb.shape) == tuple(GOLD[name + ".rgb_shape"])
    assert np.abs(rgb.reshape(-1)[GOLD[name + ".rgb_idx"]] - GOLD[name + ".rgb_val"]).max() <= 2e-6  # float: tolerance 2e-6 abs
    assert abs(rgb.mean(dtype=np.float64) - GOLD[name + ".rgb_stats"][0]) < 1e-6
    assert np.array_equal(o["intrinsics_b44t"], GOLD[name + ".intrinsics_b44t"])
    assert np.array_equal(o["track_2d_pointquerries_bn3"], GOLD[name + ".queries"])
    assert np.array_equal(o["track_2d_pointlabels_bn"], GOLD[name + ".labels"])
    assert int(o["ori_video_len"]) == int(GOLD[name + ".ori_video_len"])


def test_oracle_matches_pillow_directly():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for (h, w), (pw, ph) in [((37, 53), (224, 224)), ((300, 500), (224, 224)), ((224, 224), (224, 224)), ((224, 301), (224, 224)),
                             ((90, 90), (17, 200)), ((1, 9), (5, 3))]:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((pw, ph), resample=Image.Resampling.BILINEAR))
        assert np.array_equal(po.pil_resize_u8(img, pw, ph), ref), ((h, w), (pw, ph))


def test_mirror_pad_indices_and_edges():
    assert po.mirror_pad_indices(1, 16) == [0] * 16
    assert po.mirror_pad_indices(20, 32)[:32] == list(range(20)) + list(range(18, 6, -1))
    assert len(po.mirror_pad_indices(3, 16)) == 17  # 3 -> 5 -> 9 -> 17
    assert po.mirror_pad_indices(16, 16) == list(range(16))
    assert po.mirror_pad_indices(2, 5) == [0, 1, 0, 1, 0]


def test_abi_pil_coeffs_equal_oracle_tables():
    from l4p_amd import _lib

    lib = _lib.load()  # loads without a GPU; l4p_pil_coeffs is a host function
    for n_in, n_out in [(854, 224), (224, 854), (480, 224), (224, 480), (135, 298), (241, 224), (224, 224), (7, 3), (3, 7),
                        (1080, 224), (1920, 224), (1, 1)]:
        ks = C.c_int(0)
        assert lib.l4p_pil_coeffs(n_in, n_out, None, None, 0, C.byref(ks)) == 0
        b = np.empty((n_out, 2), dtype=np.int32)
        k = np.empty((n_out, ks.value), dtype=np.int32)
        assert lib.l4p_pil_coeffs(n_in, n_out, b.ctypes.data, k.ctypes.data, k.size, C.byref(ks)) == 0
        ob, ok, oks = po.pil_coeffs(n_in, n_out)
        assert oks == ks.value and np.array_equal(ob, b) and np.array_equal(ok, k), (n_in, n_out)
    # error behaviour: a too-small coefficient buffer and non-positive sizes are refused
    assert lib.l4p_pil_coeffs(10, 5, b.ctypes.data, k.ctypes.data, 1, C.byref(ks)) != 0
    assert lib.l4p_pil_coeffs(0, 5, None, None, 0, C.byref(ks)) != 0


def test_abi_resize_index_table_equals_oracle():
    from l4p_amd import _lib

    lib = _lib.load()
    for n_in, res, crop0, n_out in [(854, 224, 0, 224), (480, 224, 0, 224), (120, 224, 0, 224), (241, 224, 0, 224),
                                    (135, 298, 37, 224), (224, 224, 0, 224), (1920, 224, 0, 224), (5, 7, 2, 3)]:
        i0, i1 = np.empty(n_out, dtype=np.int32), np.empty(n_out, dtype=np.int32)
        lam = np.empty(n_out, dtype=np.float32)
        assert lib.l4p_resize_index_table(n_in, res, crop0, n_out, i0.ctypes.data, i1.ctypes.data, lam.ctypes.data) == 0
        o0, o1, _, ol = po.interp_axis(n_in, res)
        sl = slice(crop0, crop0 + n_out)
        assert np.array_equal(i0, o0[sl]) and np.array_equal(i1, o1[sl]) and np.array_equal(lam, ol[sl]), (n_in, res)
    assert lib.l4p_resize_index_table(10, 8, 4, 8, i0.ctypes.data, i1.ctypes.data, lam.ctypes.data) != 0  # crop outside the axis


def test_product_path_does_not_import_the_oracle():
    src = open(os.path.join(ROOT, "l4p_amd", "data", "video_dataset.py")).read()
    assert "oracle" not in src.replace("never imports oracle/", "")
