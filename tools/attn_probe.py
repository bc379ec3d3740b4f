"""Run only the fused attention kernel (encoder shape) a few times — target for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from l4p_amd import ops

B = int(os.environ.get("MB_B", "1"))
S, H, Dh = 2048, 16, 88
q = torch.randn(B * S, H * 96, device="cuda").bfloat16()
kt = torch.randn(B * S * H * 96, device="cuda").bfloat16()
vt = torch.randn(B, H, 96, S, device="cuda").bfloat16()
for _ in range(5):
    o = ops.attention(q, kt, vt, Dh)
torch.cuda.synchronize()
print("done", float(o.float().abs().mean()))
