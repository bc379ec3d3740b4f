"""Seeded synthetic inputs + sampling helpers shared by tools/gen_golden.py and the parity tests.
(randn seed 1234 clip, fx=fy=224 / cx=cy=112 intrinsics, grid point queries.)"""
import torch

from l4p_amd.data.synthetic import grid_queries, synthetic_video  # noqa: F401  (shared with bench.py / demo.py)

QUERY_TIMES = [0, 0, 3, 10, 0, 18, 7, 25, 1, 12, 0, 21]


def make_batch(T: int, nq: int, seed: int = 1234):
    g = torch.Generator().manual_seed(seed)
    rgb = torch.randn([1, 3, T, 224, 224], generator=g, dtype=torch.float32)
    K = torch.eye(4, dtype=torch.float32)
    K[0, 0] = K[1, 1] = 224.0
    K[0, 2] = K[1, 2] = 112.0
    K = K[None, :, :, None].repeat(1, 1, 1, T)
    q = torch.zeros(1, nq, 3)
    for i in range(nq):
        t = min(QUERY_TIMES[i % len(QUERY_TIMES)], T - 2)
        q[0, i] = torch.tensor([t + 0.5, 14.0 + 28.0 * (i % 8) + 0.5, 14.0 + 28.0 * ((3 * i + 1) % 8) + 0.5])
    return {
        "rgb_b3thw": rgb,
        "intrinsics_b44t": K,
        "track_2d_pointquerries_bn3": q,
        "track_2d_pointlabels_bn": torch.ones(1, nq),
    }


def sample_indices(numel: int, n: int = 4096) -> torch.Tensor:
    if numel <= n:
        return torch.arange(numel)
    return torch.linspace(0, numel - 1, n, dtype=torch.float64).round().long()


# ---- clip preparation: seeded synthetic "decoded video" frames and the fixture cases ----------
PREPROCESS_CASES = {
    # up-scale 120x160 -> 224x224 and back (the blur is nearly the identity), mirror-pad 20 -> 39 frames, crop to 32
    "small_up": dict(seed=11, T=20, H=120, W=160, crop_size=(32, 224, 224), resize_size=(224, 224), max_frames=192,
                     stride=1, spacing=0.04),
    # down-scale 270x480 (support 1.2 / 2.1 taps), stride 2, crop_size None -> ceil(max(T,16)/8)*8 frames
    "down_stride": dict(seed=12, T=21, H=270, W=480, crop_size=None, resize_size=(224, 224), max_frames=192, stride=2,
                        spacing=0.1),
    # max_frames cuts the video (the reference keeps max_frames - 1 frames), odd sizes, non-square resize + centre crop
    "cut_nonsquare": dict(seed=13, T=12, H=135, W=241, crop_size=(16, 224, 224), resize_size=(298, 224), max_frames=10,
                          stride=1, spacing=0.25),
    # a single frame is repeated (l4p_dataset_mini.py:553-554)
    "single_frame": dict(seed=14, T=1, H=64, W=96, crop_size=(16, 224, 224), resize_size=(224, 224), max_frames=192,
                         stride=1, spacing=0.5),
}


def single_window_batch():
    """One 16-frame clip for the NON-windowed entry (always_use_windowed_version=False): queries at mixed times and point
    labels 0 / 1 / 2 given by the caller (the single-window forward must use them as they are)."""
    b = make_batch(16, 6)
    b["track_2d_pointlabels_bn"] = torch.tensor([[1.0, 0.0, 2.0, 1.0, 2.0, 0.0]])
    return b


def synthetic_rays(B=2, T=4, h=16, w=16, seed=5, noise=2e-3):
    """Pluecker ray maps [B,6,T,h,w] of cameras with ONE ray-grid K per batch item, rotations / centres per frame."""
    g = torch.Generator().manual_seed(seed)
    Ks, rays = [], torch.zeros(B, 6, T, h, w)
    j, i = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    pix = torch.stack([i, j, torch.ones_like(i)], dim=-1).reshape(-1, 3)
    for b in range(B):
        K = torch.tensor([[13.0 + b, 0.2, 7.3], [0.0, 14.5 - b, 7.9], [0.0, 0.0, 1.0]])
        Ks.append(K)
        for t in range(T):
            R = torch.linalg.qr(torch.eye(3) + 0.15 * torch.randn(3, 3, generator=g)).Q
            if torch.linalg.det(R) < 0:
                R = -R
            c = torch.randn(3, generator=g)
            d_cam = (torch.inverse(K) @ pix.T).T
            d_cam = d_cam / d_cam.norm(dim=-1, keepdim=True)
            d = d_cam @ R  # world-frame directions of cam_T_world rotation R: d_world = R^T d_cam
            d = d + noise * torch.randn(d.shape, generator=g)
            m = torch.cross(c.expand_as(d), d, dim=-1)
            rays[b, :3, t] = d.T.reshape(3, h, w)
            rays[b, 3:, t] = m.T.reshape(3, h, w)
    return rays, Ks
