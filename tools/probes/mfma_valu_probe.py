"""Which MFMA forms on one wave disturb which VALU forms on ANOTHER wave of the same SIMD?  (round-5 root cause of the "stream race")

Victim: valu_victim_kernel<form> (tools/probes/race_victim.hip) - one VALU instruction repeated on fixed inputs, results compared with
a reference launch made while nothing else runs.  Aggressor: aggressor2_kernel<kind> - one MFMA form in a loop, chip-filling, on a
second stream.  Prints, per (aggressor, victim form): iterations with a wrong result, split by quarter wave and by result dword.

  python tools/probes/mfma_valu_probe.py [--aggr 0,1,...] [--forms 0,1,...] [--iters 200]
"""
import argparse
import ctypes as C
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
AGGR = ["16x16x32 bf16 (chain)", "16x16x32 bf16 (4 accumulators)", "16x16x32 f16", "16x16x16 bf16_1k", "16x16x4 f32", "32x32x2 f32",
        "16x16x32 fp8", "32x32x16 f16", "32x32x16 bf16", "4x4x4 f16", "32x32x8 bf16_1k"]
FORMS = ["v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]",
         "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]", "v_pk_mov_b32 op_sel:[1,0]", "v_mul_f32", "v_fma_f32",
         "v_pk_fma_f16", "v_fma_f64", "v_mad_u64_u32", "v_lshl_add_u64", "v_cvt_pk_bf16_f32", "v_pk_add_f32",
         "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,1]", "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0]", "v_pk_fma_f32 op_sel_hi:[1,0,1]",
         "v_pk_mul_f16 op_sel:[0,1] op_sel_hi:[1,0]", "v_add_f32_dpp row_shr:1", "v_permlane32_swap_b32", "v_mul_f64",
         "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]",
         "v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1]", "v_pk_mov_b32 op_sel:[0,1]", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1]",
         "v_pk_mul_f32 d, a, a op_sel:[0,1] op_sel_hi:[1,0] (src0 = src1)", "v_pk_mul_f32 d, b, a op_sel:[1,0] op_sel_hi:[0,1] (operands exchanged)"]
ap = argparse.ArgumentParser()
ap.add_argument("--aggr", default=",".join(str(i) for i in range(len(AGGR))))
ap.add_argument("--forms", default=",".join(str(i) for i in range(len(FORMS)) if i != 20))  # (20: the harness's reference launch does
#                                                              not reproduce the swap's second destination; not part of the finding)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--aggr-iters", type=int, default=4000)
args = ap.parse_args()

lib = C.CDLL(os.path.join(ROOT, "tools", "probes", "librace_victim.so"))
lib.race_aggr2_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.race_valu_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
dev = torch.device("cuda")
print("device:", torch.cuda.get_device_properties(0).gcnArchName)
n = args.n
sink = torch.zeros(256, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def stream():
    return torch.cuda.current_stream().cuda_stream


for form in [int(x) for x in args.forms.split(",")]:
    ref = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    assert lib.race_valu_launch(stream(), form, ref.data_ptr(), None, n, 1, 0) == 0
    torch.cuda.synchronize()
    row = []
    for kind in [-1] + [int(x) for x in args.aggr.split(",")]:
        cnts = []
        if kind >= 0:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(args.launches * 3):
                    assert lib.race_aggr2_launch(stream(), kind, 2048, args.aggr_iters, sink.data_ptr()) == 0
        for _ in range(args.launches):
            c = torch.zeros(2 * n, dtype=torch.int32, device=dev)
            assert lib.race_valu_launch(stream(), form, ref.data_ptr(), c.data_ptr(), n, args.iters, 1) == 0
            cnts.append(c)
        main.wait_stream(side)
        torch.cuda.synchronize()
        tot = torch.stack(cnts).sum(0).view(n, 2)
        bad, badlo = tot[:, 0], tot[:, 1]
        q = [int(bad.view(-1, 4, 16)[:, k].sum()) for k in range(4)]
        row.append((kind, int(bad.sum()), int(badlo.sum()), q))
    total = args.launches * args.iters * n
    print(f"form {form:2d} {FORMS[form]}")
    for kind, b, bl, q in row:
        if b or kind < 0:
            print(f"     beside {'nothing' if kind < 0 else 'MFMA ' + AGGR[kind]:38s}: wrong results {b:9d} of {total} "
                  f"({b / total:.2e}); low dword wrong in {bl}; by quarter wave (lanes 0-15, 16-31, 32-47, 48-63): {q}")
