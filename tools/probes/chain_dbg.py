import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from l4p_amd import _lib
from l4p_amd._lib import L4P_F32
from l4p_amd.ops import _p, _stream
lib = _lib.load(); dt = L4P_F32; td = torch.float32
P, Cc, N = 96, 1408, 3; M = N * P
g = torch.Generator().manual_seed(21); r = lambda *s: torch.randn(*s, generator=g)
xs, pos = r(P, Cc).cuda(), r(P, Cc).cuda()
d0, d1 = (0.5 * r(M, Cc)).cuda(), (0.5 * r(M, Cc)).cuda()
g0, b0, g1, b1 = (1 + 0.1 * r(Cc)).cuda(), (0.1 * r(Cc)).cuda(), (1 + 0.1 * r(Cc)).cuda(), (0.1 * r(Cc)).cuda()
e = lambda: torch.empty(M, Cc, device="cuda")
y0, kT0, kP0, kT1, kP1, y1 = e(), e(), e(), e(), e(), e()
lib.l4p_layernorm_res(_stream(), dt, _p(xs), P, _p(d0), _p(g0), _p(b0), 1e-5, _p(kT0), _p(y0), M, Cc, _p(pos), P, _p(kP0), None, 1, 0, None)
lib.l4p_layernorm_res(_stream(), dt, _p(y0), 0, _p(d1), _p(g1), _p(b1), 1e-5, _p(kT1), _p(y1), M, Cc, _p(pos), P, _p(kP1), None, 1, 0, None)
st = torch.empty(M, 2, device="cuda"); cT0, cP0, cT1, cP1, c1 = e(), e(), e(), e(), e()
lib.l4p_layernorm_res(_stream(), dt, _p(xs), P, _p(d0), _p(g0), _p(b0), 1e-5, _p(cT0), None, M, Cc, _p(pos), P, _p(cP0), None, 1, 0, _p(st))
zero = torch.zeros(M, Cc, device="cuda")
# chain with zero second update and identity second norm -> exposes the re-derived y0 through out_f32? (LN again) - instead compare via d1 path
lib.l4p_layernorm_chain(_stream(), dt, _p(xs), P, _p(d0), _p(st), _p(g0), _p(b0), _p(d1), _p(g1), _p(b1), 1e-5, _p(cT1), _p(c1), M, Cc, _p(pos), P, _p(cP1))
torch.cuda.synchronize()
for n, a, b in (("kT1", kT1, cT1), ("kP1", kP1, cP1), ("y1", y1, c1)):
    d = (a - b).abs(); print(n, "max diff", float(d.max()), "differing", int((d > 0).sum()), "of", d.numel(), "rows", int((d > 0).any(dim=1).sum()))
# recompute y0 on host from stats, compare with stored y0
x0 = xs.repeat(N, 1) + d0
y0h = torch.addcmul(b0, (x0 - st[:, :1]) * st[:, 1:], g0)   # not fma-exact, just magnitude
print("y0 host vs stored", float((y0h - y0).abs().max()))
d = (y1 - c1).abs()
rows = (d > 0).any(dim=1).nonzero().flatten().tolist()
print("rows", rows)
r0 = rows[0]; cols = (d[r0] > 0).nonzero().flatten().tolist(); print("row", r0, "ncols", len(cols), cols[:10])
# stats check vs torch
x0 = xs.repeat(N, 1) + d0
print("mean diff", float((st[:, 0] - x0.mean(dim=1)).abs().max()))
# second-stage statistics from both paths
v1 = y0 + d1
print("row", r0, "mean(v1)", float(v1[r0].double().mean()), "y1 row mean", float(y1[r0].double().mean()), float(c1[r0].double().mean()))
