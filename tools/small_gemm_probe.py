"""Timing of the tracker's tiny GEMMs (M = 64..384) with and without split-K (tuning aid)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from l4p_amd import _lib, ops
from l4p_amd._lib import GemmDesc

def run(M, N, K, sk):
    dev = "cuda"
    a = torch.randn(M, K, device=dev).bfloat16()
    w = ops.pad_rows(torch.randn(N, K, device=dev).bfloat16() * K ** -0.5)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    part = torch.empty(max(sk, 1) * M * N, device=dev)
    d = GemmDesc()
    d.A, d.lda, d.W, d.ldw = a.data_ptr(), K, w.data_ptr(), K
    d.M, d.N, d.K = M, N, K
    d.bias = bias.data_ptr(); d.out_T = out.data_ptr(); d.ldc = N
    d.splitk = sk; d.partial = part.data_ptr()
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5): _lib.check(lib.l4p_gemm(st, 0, C.byref(d)))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): lib.l4p_gemm(st, 0, C.byref(d))
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 50 * 1e3, out.float()

for (M, N, K) in [(384, 1408, 1408), (384, 704, 1408), (64, 1408, 1408), (384, 1408, 2048), (384, 2048, 1408), (384, 1408, 704), (64, 176, 1408), (2048, 704, 1408)]:
    base, ref = run(M, N, K, 1)
    line = f"M={M} N={N} K={K}: sk1 {base:6.1f} us"
    for sk in (2, 4, 8):
        t, o = run(M, N, K, sk)
        line += f" | sk{sk} {t:6.1f} us (maxdiff {float((o-ref).abs().max()):.3g})"
    print(line)
