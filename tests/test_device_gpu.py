"""The engine's device contract: l4p_create leaves the caller's current device alone, and a model pinned to a GPU runs there
whatever the current device is (round-3 advisor finding).  The second test needs two GPUs and is skipped on the 1-GPU pool."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd import _lib
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch


def test_l4p_create_keeps_current_device_and_rejects_unknown_devices(dev):
    lib = _lib.load()
    before = torch.cuda.current_device()
    h = C.c_void_p()
    assert lib.l4p_create(torch.cuda.device_count() - 1, _lib.L4P_F32, C.byref(h)) == 0
    assert torch.cuda.current_device() == before
    lib.l4p_destroy(h)
    assert lib.l4p_create(torch.cuda.device_count(), _lib.L4P_F32, C.byref(h)) != 0
    assert b"no such device" in lib.l4p_last_error()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_model_on_a_non_current_device(dev):
    from tests.test_encoder_dpt_gpu import build

    cfg = ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    tasks = ["depth", "track_2d"]
    batch = make_batch(16, 4)
    torch.cuda.set_device(0)
    m0 = build(cfg, sd, "32-true")
    with torch.no_grad():
        ref = m0.forward({k: v.clone() for k, v in batch.items()}, tasks)
    with torch.cuda.device(1):
        m1 = build(cfg, sd, "32-true")  # weights arrive while cuda:1 is current: the model lives there
    assert m1.l4p_model.device == torch.device("cuda", 1) and torch.cuda.current_device() == 0
    with torch.no_grad():
        out = m1.forward({k: v.clone() for k, v in batch.items()}, tasks)  # called with cuda:0 current
    assert torch.cuda.current_device() == 0
    for k, v in ref.items():
        if torch.is_tensor(v):
            assert out[k].device == torch.device("cuda", 1), k
            assert torch.equal(out[k].cpu(), v.cpu()), k
