import sys; sys.path.insert(0,'/root/repo')
import torch
from l4p_amd import ops
from tests.test_kernels_gpu import _attn_inputs
from l4p_amd._lib import L4P_BF16
def run(B,S,H,Dh,peak):
    g = torch.Generator().manual_seed(5)
    q4 = torch.randn(B, S, H, ops.DP, generator=g); k4 = torch.randn(B, S, H, ops.DP, generator=g); v4 = torch.randn(B, S, H, ops.DP, generator=g)
    for t in (q4,k4,v4): t[..., Dh:] = 0
    if peak:
        k4[0, 300:310] *= 6.0; q4[0, 17] *= 8.0; k4[0, :64] *= 0.01
    q, kt, vt, qf, kf, vf = _attn_inputs(q4, k4, v4, L4P_BF16)
    qh, kh, vh = (t.permute(0, 2, 1, 3).double() for t in (qf, kf, vf))
    attn = torch.softmax((qh * Dh ** -0.5) @ kh.transpose(-2, -1), dim=-1)
    ref = (attn @ vh)[..., :Dh].transpose(1, 2).reshape(B * S, H * Dh).float()
    out = ops.attention(q, kt, vt, Dh).float().cpu()
    e=(out-ref).abs()
    rows=e.max(dim=1).values
    print(B,S,H,Dh,peak,'max',float(e.max()),'scale',float(ref.abs().max()),'relL2',float((out-ref).norm()/ref.norm()),'worst rows',rows.topk(5).indices.tolist(), rows.topk(5).values.tolist(), 'nan', int(torch.isnan(out).sum()))
run(1,512,2,88,True); run(1,512,2,88,False); run(4,2048,16,88,True); run(1,2048,16,88,True); run(2,512,8,64,True)
