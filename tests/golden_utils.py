"""Seeded synthetic inputs + sampling helpers shared by tools/gen_golden.py and the parity tests.
(randn seed 1234 clip, fx=fy=224 / cx=cy=112 intrinsics, grid point queries.)"""
import torch

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
