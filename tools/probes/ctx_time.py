"""l4p_t2i_context alone (the context product of the folded token -> image attention): time per launch for a few track counts, with
the eight-stage ring on / off.  usage: python tools/probes/ctx_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from l4p_amd import _lib
from l4p_amd._lib import L4P_BF16
from l4p_amd.ops import _p, _stream

lib = _lib.load()
Cc, heads, tokens = 1408, 8, 6
HT = heads * tokens
for N, P in ((1, 2048), (8, 2048), (16, 2048), (23, 2048), (64, 2048)):
    sc = (3.0 * torch.randn(N * P, HT)).cuda()
    keys = torch.randn(N * P, Cc).bfloat16().cuda()
    nsp = (P + 255) // 256
    pr = torch.empty(N * P, HT, dtype=torch.bfloat16, device="cuda")
    st = torch.empty(N * nsp, 2 * HT, device="cuda")
    _lib.check(lib.l4p_t2i_probs(_stream(), L4P_BF16, _p(sc), HT, _p(pr), _p(st), N, P, HT), "probs")
    Rg = (tokens * N + 127) // 128 * 128
    cx = torch.zeros(heads * Rg, Cc, dtype=torch.bfloat16, device="cuda")
    line = f"N={N:3d} P={P:4d} ({(Cc // 128) * N} workgroups):"
    for deep in (0, 1):
        _lib.set_knob("track_deep", deep)
        for _ in range(5):
            lib.l4p_t2i_context(_stream(), L4P_BF16, _p(pr), _p(st), _p(keys), _p(cx), N, P, Cc, heads, tokens, Rg, P)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100):
            lib.l4p_t2i_context(_stream(), L4P_BF16, _p(pr), _p(st), _p(keys), _p(cx), N, P, Cc, heads, tokens, Rg, P)
        e.record()
        torch.cuda.synchronize()
        line += f"  track_deep={deep}: {s.elapsed_time(e) * 10:.1f} us"
    print(line)
