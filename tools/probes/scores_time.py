"""The folded score product of a query shard (rows [g P, (g + 1) P) of the keys x track g's 48 folded rows) on the staged row-grouped
kernel and on the one-wave kernel: time per launch.  (Measured with the launcher's 512-block limit of gemm_is_skinny_grouped lifted:
8 tracks 12.4 us staged against 32.7 us one-wave, bit-identical; the limit stays, so both columns now show the staged kernel.)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from l4p_amd import _lib
from l4p_amd._lib import GemmDesc, L4P_BF16, EPI_DENSE
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
P, HT, K = 2048, 48, 1408
for G in (4, 8, 16, 64):
    a = torch.randn(G * P, K, device="cuda").bfloat16()
    w = (torch.randn(G * HT + 128, K, device="cuda") * K ** -0.5).bfloat16()
    o = torch.zeros(G * P, HT, device="cuda")
    d = GemmDesc()
    d.A, d.lda, d.W, d.ldw, d.M, d.N, d.K = a.data_ptr(), K, w.data_ptr(), K, G * P, HT, K
    d.out_f32, d.ldc, d.epi = o.data_ptr(), HT, EPI_DENSE
    d.w_gr, d.w_gs, d.b_gs = P, HT * K, 0
    line = f"tracks={G:3d} (M={G * P}):"
    outs = []
    for sk in (0, 1):
        _lib.set_knob("gemm_skinny", sk)
        for _ in range(5): lib.l4p_gemm(st, L4P_BF16, C.byref(d))
        torch.cuda.synchronize()
        outs.append(o.clone())
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100): lib.l4p_gemm(st, L4P_BF16, C.byref(d))
        e.record(); torch.cuda.synchronize()
        line += f"  skinny={sk}: {s.elapsed_time(e) * 10:.1f} us"
    print(line, " equal:", bool(torch.equal(outs[0], outs[1])))
