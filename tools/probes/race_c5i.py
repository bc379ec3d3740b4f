"""Victim: the seam alignment's pointmap kernel (main stream, fixed inputs, fresh outputs).  Aggressors, one at a time on a side
stream: the tracker's kernel families.  Which one makes the victim's output differ?  (round-4 race diagnosis)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "race_c5c.py")).read().split("def once():")[0])
from l4p_amd import ops
from l4p_amd._lib import ACT_NONE, L4P_BF16

bf = torch.bfloat16


def victim():
    a = torch.empty(n, 3, device=dev)
    _lib.check(lib.l4p_point_map_samples(_stream(), _p(depth), _p(K), _p(P), _p(a), F, H, W, ratio, seed), "p")
    return a


ref = victim().clone()
torch.cuda.synchronize()
side = torch.cuda.Stream()
Cc, Pk, Nq, Dh, heads = 704, 1568, 2, 352, 8
xa = torch.randn(Nq * Pk, Cc, device=dev).to(bf)
w_big = ops.pad_rows(torch.randn(Cc, Cc, device=dev).to(bf) * 0.03, 256)
xs = torch.randn(6 * Nq, Cc, device=dev).to(bf)
x32 = torch.randn(Nq * Pk, Cc, device=dev)
g = torch.ones(Cc, device=dev)
b = torch.zeros(Cc, device=dev)
xbig = torch.randn(65536, 1408, device=dev).to(bf)
wbig = ops.pad_rows(torch.randn(1408, 1408, device=dev).to(bf) * 0.03, 256)
tq = torch.randn(6 * Nq, Dh, device=dev).to(bf)
tk = torch.randn(Nq * Pk, Dh, device=dev).to(bf)
tv = torch.randn(Nq * Pk, Dh, device=dev).to(bf)
to = torch.empty(6 * Nq, Dh, device=dev, dtype=bf)
io = torch.empty(Nq * Pk, Dh, device=dev, dtype=bf)
masks = torch.randn(Nq, 3, 16, 56, 56, device=dev)
traj = torch.empty(Nq, 2, 16, device=dev)
vis = torch.empty(Nq, 16, device=dev)
dep = torch.empty(Nq, 16, device=dev)
hist = torch.empty(Nq * Pk, Cc, device=dev)
tokrow = torch.randn(Cc, device=dev)


def small_attn(kind, q, k, v, o, P_):
    _lib.check(lib.l4p_small_attn(_stream(), L4P_BF16, kind, _p(q), _p(k), _p(v), _p(o), Nq, P_, Dh, heads), "sa")


AGG = {
    "nothing": lambda: None,
    "gemm M3136 N704 K704 (key-side projection, mini)": lambda: ops.gemm(xa, w_big, Cc),
    "gemm M12 N704 K704 (token-side, 4-stage)": lambda: ops.gemm(xs, w_big, Cc),
    "gemm M65536 N1408 K1408 (8-phase)": lambda: ops.gemm(xbig, wbig, 1408),
    "layernorm f32 -> bf16 (3136 x 704)": lambda: ops.layernorm(x32, g, b, 1e-5, L4P_BF16),
    "small_attn kind 1 (tokens -> image)": lambda: small_attn(1, tq, tk, tv, to, Pk),
    "small_attn kind 2 (image -> tokens)": lambda: small_attn(2, tk, tq, tq, io, Pk),
    "track_readout": lambda: _lib.check(lib.l4p_track_readout(_stream(), _p(masks), _p(traj), _p(vis), _p(dep), Nq, 16, 56, 56, 224, 224), "ro"),
    "fill_rows": lambda: _lib.check(lib.l4p_fill_rows(_stream(), _p(hist), _p(tokrow), Nq * Pk, Cc, Nq * Pk, 0, 0), "fr"),
}
for name, fn in AGG.items():
    fn()
    torch.cuda.synchronize()
    bad = runs = 0
    for rep in range(6):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            keep = [fn() for _ in range(150)]
        res = [victim() for _ in range(80)]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(r, ref)) for r in res)
        runs += len(res)
        del keep, res
    print(f"{name:55s} victim mismatches {bad} / {runs}")
