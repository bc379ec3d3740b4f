"""Times of the GEMM shapes an absorbed image -> token attention would run (keys x K'^T: N = 48, K = 1408; P x V': N = 1408, K = 64)
next to the ones it would replace (i2t.q: N = 704, K = 1408; i2t.out: N = 1408, K = 704), 64 tracks x 2048 keys."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from l4p_amd import ops

M = 131072
bf = torch.bfloat16


def t(fn, n=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


x1408 = torch.randn(M, 1408, device="cuda").to(bf)
x704 = torch.randn(M, 704, device="cuda").to(bf)
x64 = torch.randn(M, 64, device="cuda").to(bf)
for name, a, N, K in (("i2t.q   N704 K1408", x1408, 704, 1408), ("i2t.out N1408 K704", x704, 1408, 704),
                      ("scores  N48 K1408", x1408, 48, 1408), ("scores  N64 K1408", x1408, 64, 1408), ("scores N128 K1408 (t2i+i2t together)", x1408, 128, 1408),
                      ("delta   N1408 K64", x64, 1408, 64)):
    w = ops.pad_rows(torch.randn(N, K, device="cuda").to(bf) * K ** -0.5, 256)
    bias = torch.zeros(N, device="cuda")
    us = t(lambda: ops.gemm(a, w, N, bias=bias))
    us32 = t(lambda: ops.gemm(a, w, N, bias=bias, out_f32=True, out_T=False)) if N <= 128 else float("nan")
    print(f"{name:40s} {us:8.1f} us (bf16 out)  {us32:8.1f} us (f32 out)")
