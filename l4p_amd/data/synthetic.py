"""Seeded synthetic inputs of the benchmark and demo workloads: the 8x8 grid of track queries and a
deterministic uint8 "decoded video".  Lives in the package so that bench.py and demo/demo.py do not depend on tests/."""
import torch


def grid_queries(nq: int) -> torch.Tensor:
    """The benchmark's track queries (bench.py): an 8x8 grid x, y in {14 + 28 i} + 0.5 at t = 0.5 -> [1,nq,3]."""
    q = torch.zeros(1, nq, 3)
    for i in range(nq):
        q[0, i] = torch.tensor([0.5, 14.0 + 28.0 * (i % 8) + 0.5, 14.0 + 28.0 * ((i // 8) % 8) + 0.5])
    return q


def synthetic_video(seed: int, T: int, H: int, W: int):
    """uint8 frames [T,H,W,3]: smooth moving gradients + blocks + noise, so every filter tap matters."""
    import numpy as np

    rng = np.random.default_rng(seed)
    t = np.arange(T)[:, None, None, None]
    y = np.arange(H)[None, :, None, None]
    x = np.arange(W)[None, None, :, None]
    c = np.arange(3)[None, None, None, :]
    v = 127 + 90 * np.sin(0.07 * x + 0.3 * t + c) * np.cos(0.05 * y - 0.2 * t) + 40 * (((x // 8 + y // 8 + t) % 2) - 0.5)
    v = v + rng.normal(0, 12, size=(T, H, W, 3))
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)
