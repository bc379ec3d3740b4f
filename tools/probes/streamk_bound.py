"""Upper bound of what balancing the 256x256-tile GEMMs of the batch-4 encoder over all 256 CUs could give: each real shape
next to a proxy with the same per-CU work but a whole number of rounds (random bf16 data, bias + bf16 store)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from l4p_amd import ops


def t(M, N, K, n=100):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.randn(M, K, device="cuda", generator=g)).bfloat16()
    w = (torch.randn((N + 127) // 128 * 128, K, device="cuda", generator=g) * 0.03).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm(a, w, N, bias=b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, N, bias=b, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"M{M} N{N} K{K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF/s  tiles {((M + 255) // 256) * ((N + 255) // 256)}", flush=True)
    return us


for rep in range(2):
    print("fc2   real"); t(8192, 1408, 6144)
    print("fc2   proxy: 256 tiles, K x 0.75"); t(8192, 2048, 4608)
    print("proj  real"); t(8192, 1408, 1408)
    print("proj  proxy"); t(8192, 2048, 1056)
    print("qkv   real (dense epilogue)"); t(8192, 4608, 1408)
    print("qkv   proxy: 512 tiles, K x 1.125"); t(8192, 4096, 1584)
    print("fc1   real (3 whole rounds)"); t(8192, 6144, 1408)
