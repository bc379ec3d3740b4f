"""Time the fused attention kernel alone on the encoder shape (random data; --zeros: all-zero operands): us per launch and TF/s
of useful FLOPs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from l4p_amd import ops

S, H, Dh = 2048, 16, 88
ZERO = "--zeros" in sys.argv  # all-zero operands: the same instruction stream with (almost) no switching activity in the matrix pipe
for B in (4, 1, 8):
    g = torch.Generator(device="cuda").manual_seed(B)
    q = torch.randn(B * S, H * 96, device="cuda", generator=g).bfloat16()
    kt = torch.randn(B * S * H * 96, device="cuda", generator=g).bfloat16()
    vt = torch.randn(B, H, 96, S, device="cuda", generator=g).bfloat16()
    if ZERO:
        q.zero_(), kt.zero_(), vt.zero_()
    for _ in range(10):
        o = ops.attention(q, kt, vt, Dh)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    a.record()
    for _ in range(n):
        o = ops.attention(q, kt, vt, Dh)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    fl = 4.0 * S * S * Dh * H * B
    print(f"B={B}: {us:.1f} us  {fl / us / 1e6:.0f} TF/s useful ({fl / us / 1e6 / 2500 * 100:.1f} % of 2.5 PF)", flush=True)
