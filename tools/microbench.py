"""Kernel microbenchmarks on one MI355X: encoder GEMM shapes + fused attention. Prints TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from l4p_amd import ops
from l4p_amd._lib import L4P_BF16, L4P_F32, ACT_GELU


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    modes = [L4P_BF16] + ([L4P_F32] if "--f32" in sys.argv else [])
    B = int(os.environ.get("MB_B", "1"))
    for mode in modes:
        td = ops.torch_dtype(mode)
        print("mode", "bf16" if mode == L4P_BF16 else "f32", "B", B)
        M = 2048 * B
        for (N, K, name) in [(4608, 1408, "qkv"), (1408, 1408, "proj"), (6144, 1408, "fc1"), (1408, 6144, "fc2")]:
            a = torch.randn(M, K, device="cuda").to(td)
            w = ops.pad_rows((torch.randn(N, K, device="cuda") * K ** -0.5).to(td))
            bias = torch.randn(N, device="cuda")
            out = torch.empty(M, N, device="cuda", dtype=td)
            t = timeit(lambda: ops.gemm(a, w, N, bias=bias, out=out))
            print(f"  gemm {name:5s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s")
        H, Dh, S = 16, 88, 2048
        q = torch.randn(B * S, H, 96, device="cuda")
        q[..., Dh:] = 0
        q = q.reshape(B * S, H * 96).to(td)
        kt = torch.randn(B * S * H * 96, device="cuda").to(td)
        vt = torch.randn(B, H, 96, S, device="cuda").to(td)
        t = timeit(lambda: ops.attention(q, kt, vt, Dh))
        fl = 4 * S * S * Dh * H * B
        print(f"  attention B={B} S={S} H={H} d={Dh}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s useful ({fl*96/88/t/1e12:.1f} raw)")
        # conv3d shapes of the DPT head
        for (shape, cout, name) in [((B, 16, 64, 64, 256), 256, "rn1"), ((B, 16, 128, 128, 256), 128, "head1"),
                                    ((B, 16, 224, 224, 128), 128, "head2")]:
            x = torch.randn(shape, device="cuda").to(td)
            cin = shape[-1]
            w = ops.pad_rows((torch.randn(cout, 27 * cin, device="cuda") * (27 * cin) ** -0.5).to(td))
            bias = torch.randn(cout, device="cuda")
            t = timeit(lambda: ops.conv3d_k3(x, w, cout, bias=bias), iters=5, warm=1)
            m = shape[0] * shape[1] * shape[2] * shape[3]
            print(f"  conv3 {name:5s} M={m} Cin={cin} Cout={cout}: {t*1e6:8.1f} us  {2*m*cout*27*cin/t/1e12:7.1f} TF/s")


if __name__ == "__main__":
    main()
