"""The demo's generic-video flow end to end on the GPU (reference demo/demo.py:88-112): VideoDataset -> DataLoader(batch_size=1)
-> model.forward(batch, tasks), mini geometry.  The sample dict carries keys the heads must ignore (rgb_mean_b3111, seq_name,
ori_video_len, dummy ground-truth tracks: l4p_videomae.py:251,305 pass **data); outputs are checked against the oracle fed with
the same prepared batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from l4p_amd.data import VideoDataset
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import synthetic_video
from tests.test_encoder_dpt_gpu import build

TASKS = ["depth", "flow_2d_backward", "dyn_mask", "track_2d"]  # the task list of the demo's generic-video case (demo.py:82,99)


def test_video_dataset_to_model_forward(dev):
    from oracle.l4p_oracle import OracleModel

    cfg = ModelCfg.mini()
    sd = seeded_state_dict(cfg)
    model = build(cfg, sd, "32-true")
    frames = synthetic_video(31, 20, 96, 128)  # 20 frames -> mirror-padded, cropped to 24 = 2 windows
    ds = VideoDataset(video_paths=["videos/clip.mp4"], crop_size=(24, 224, 224), estimation_directions=[1],
                      track_2d_querry_sampling_spacing=0.5, frames={"videos/clip.mp4": frames}, device=dev)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    batch = next(iter(loader))
    assert batch["rgb_b3thw"].shape == (1, 3, 24, 224, 224) and batch["rgb_b3thw"].is_cuda
    assert batch["track_2d_pointquerries_bn3"].shape == (1, 4, 3) and batch["seq_name"] == ["clip.mp4"]
    with torch.no_grad():
        out = model.forward(batch, TASKS)
        cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
        ref = OracleModel(sd, cfg).forward(cpu, TASKS)
    torch.cuda.synchronize()
    for key, shape in [("depth_est_b1thw", (1, 1, 24, 224, 224)), ("flow_2d_backward_est_b2thw", (1, 2, 24, 224, 224)),
                       ("dyn_mask_est_b1thw", (1, 1, 24, 224, 224)),
                       ("track_2d_traj_est_bn2t", (1, 4, 2, 24)), ("track_2d_vis_est_bn1t", (1, 4, 1, 24))]:
        y, r = out[key].float().cpu(), ref[key]
        assert tuple(y.shape) == shape == tuple(r.shape), key
        assert bool(torch.isfinite(y).all()), key
        assert (y - r).abs().max() <= 1e-3 * r.abs().max(), (key, float((y - r).abs().max() / r.abs().max()))
