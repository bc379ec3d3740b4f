"""gemm_skinny vs the staged kernel by row count (knob gemm_skinny with the launcher's M limit lifted through L4P_SKINNY_MAX_M):
time per launch for the tracker's token-side shapes.  usage: python tools/probes/skinny_m_sweep.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from l4p_amd import _lib, ops
from l4p_amd._lib import GemmDesc, L4P_BF16, EPI_DENSE

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
for (N, K) in ((1408, 1408), (704, 1408), (2048, 1408), (1408, 2048), (1408, 704)):
    for M in (48, 64, 96, 128, 192, 384):
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = ops.pad_rows((torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16(), 128)
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        d = GemmDesc()
        d.A, d.lda, d.W, d.ldw, d.M, d.N, d.K, d.out_T, d.ldc, d.epi = a.data_ptr(), K, w.data_ptr(), K, M, N, K, o.data_ptr(), N, EPI_DENSE
        line = f"M={M:4d} N={N:5d} K={K:5d}:"
        for sk in (0, 1):
            _lib.set_knob("gemm_skinny", sk)
            for _ in range(5): lib.l4p_gemm(st, L4P_BF16, C.byref(d))
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(200): lib.l4p_gemm(st, L4P_BF16, C.byref(d))
            e.record(); torch.cuda.synchronize()
            line += f"  skinny={sk}: {s.elapsed_time(e) * 5:.1f} us"
        print(line)
