"""Stream-race diagnosis, one parametrised script (replaces the round-4 race_c5*.py series).

Observation being chased (round 4): while the tracker recursion runs on its own HIP stream, the seam alignment's
pointmap kernel (csrc/umeyama.hip; main stream, FIXED inputs that nothing ever writes) computed, in ~8 % of its launches, with
zeros in one 16-byte pose row for lanes 48..63 of one wave.

  python tools/probes/race_probe.py --victim real|plain|seq [--iters N] [--table torch|hipmalloc] [--aggressor tracker|none]
                                    [--reps R] [--per-rep V] [--sync-every 0|1]

  aggressor tracker | none | synth:K (tools/probes/race_victim.hip aggressor_kernel<K>: one instruction class filling the chip on a
            second stream; K = 0 v_fma_f32, 1 v_permlane32_swap, 2 v_permlane16_swap, 3 MFMA 32x32x16, 4 v_pk_fma_f32, 5 ds_read_b64_tr_b16,
            6 VALU at s_setprio 3, 7 v_exp_f32, 8 DPP, 9 v_pk_mul_f32 op_sel, 10 MFMA 16x16x32, 11 LDS-DMA, 12 v_cvt_pk_bf16_f32)
            mfma:K (aggressor2_kernel<K>: one MFMA form; K = 0 16x16x32 bf16, 2 16x16x32 f16, 8 32x32x16 bf16, ... see the source)
  victim real   the library's own l4p_point_map_samples, output compared with a reference launch (the round-4 observation)
         plain  tools/probes/race_victim.hip victim_kernel: every row loaded three times (plain, plain again, sc0 sc1) and checked
                in the kernel; mismatches logged with lane / iteration / hardware id
         seq    victim_seq_kernel: the compiler's exact three-load sequence of the pointmap kernel, register for register
         full   victim_full_kernel: the pointmap kernel's own source (seven loads in flight, counted waits) + a second read of every
                input after a full drain: which value the kernel COMPUTED WITH differs from memory (flags: 1/2/4 pose rows 0/1/2,
                16/32/64 K rows, 128 depth, 256 computed output != recomputation, 512 stored output != computed)
         drain  the full victim's arithmetic with s_waitcnt vmcnt(0) in front of the first use of a pose value (no in-kernel check)
         nops   ... with 16 idle cycles there instead of the drain
         asm:X  the library kernel's own assembly with one hand edit (orig = none; drain = vmcnt(0) in front of the first use of a
                pose row; nops_after / nops_before = idle cycles behind / in front of the counted waits; sentinel = the pose rows'
                destination registers pre-filled with 777.0)
         canary canary_kernel: every wave parks sentinels in v0..v95, sleeps --iters x 192 cycles and dumps the registers: a changed
                register was written from outside the wave
         cnt    victim_cnt_kernel: the seven loads with sentinel-filled destinations, each copied right behind its counted
                s_waitcnt and compared with the final register (flags: bit j = load j of K0 K1 K2 depth P1 P0 P2 was not final;
                a = [early P0.x, final P0.x, early P1.y, final P1.y])
  environment knobs worth combining: PYTORCH_NO_CUDA_MEMORY_CACHING=1, AMD_SERIALIZE_KERNEL=3, GPU_MAX_HW_QUEUES=1|8,
  HSA_XNACK=0|1, L4P_TRACK_PYTHON=1 (tracker kernel by kernel from Python instead of one native call per window)
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from l4p_amd import _lib, parallel  # noqa: E402
from l4p_amd.ops import _p, _stream  # noqa: E402
from l4p_amd.weights import ModelCfg, seeded_state_dict  # noqa: E402
from tests.golden_utils import make_batch  # noqa: E402
from tests.test_encoder_dpt_gpu import build  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--victim", default="real")
ap.add_argument("--iters", type=int, default=1)
ap.add_argument("--table", default="torch")
ap.add_argument("--aggressor", default="tracker")
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--per-rep", type=int, default=80)
ap.add_argument("--queries", type=int, default=2)
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--aggr-launches", type=int, default=30)
ap.add_argument("--aggr-blocks", type=int, default=2048)
ap.add_argument("--aggr-iters", type=int, default=2000)
args = ap.parse_args()

lib = _lib.load()
vlib = C.CDLL(os.path.join(ROOT, "tools", "probes", "librace_victim.so"))
vlib.race_hip_malloc.restype = C.c_void_p
vlib.race_hip_malloc.argtypes = [C.c_size_t]
vlib.race_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for fn in (vlib.race_victim_launch, vlib.race_victim_seq_launch):
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
vlib.race_victim_full_launch.argtypes = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint]
vlib.race_victim_cnt_launch.argtypes = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_uint, C.c_int, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint]

dev = torch.device("cuda")
print("device:", torch.cuda.get_device_properties(0).gcnArchName, "| env:",
      {k: os.environ[k] for k in ("PYTORCH_NO_CUDA_MEMORY_CACHING", "AMD_SERIALIZE_KERNEL", "GPU_MAX_HW_QUEUES", "HSA_XNACK",
                                  "L4P_TRACK_PYTHON") if k in os.environ}, "| args:", vars(args))
g = torch.Generator().manual_seed(3)
H = W = 224
F = 3
ratio, seed = 10, 20250213
spf = (H * W) // ratio
n = F * spf
depth = torch.rand(F, H, W, generator=g).add_(0.5).to(dev)
K = torch.eye(4).repeat(F, 1, 1)
K[:, 0, 0] = K[:, 1, 1] = 200.0
K[:, 0, 2] = K[:, 1, 2] = 112.0
K = K.reshape(F, 16).to(dev)
P_host = torch.eye(4).repeat(F, 1, 1)
if args.victim in ("plain", "seq"):
    P_host = P_host * torch.arange(1, F + 1, dtype=torch.float32)[:, None, None]  # frame f: (f + 1) x identity
P_host = P_host.reshape(F, 16).contiguous()
if args.table == "hipmalloc":
    p_ptr = vlib.race_hip_malloc(P_host.numel() * 4)
    assert p_ptr, "hipMalloc failed"
    assert vlib.race_hip_memcpy_h2d(p_ptr, P_host.data_ptr(), P_host.numel() * 4) == 0
    P = None
else:
    P = P_host.to(dev)
    p_ptr = P.data_ptr()
torch.cuda.synchronize()

REC = np.dtype([("launch", "u4"), ("iter", "u4"), ("gid", "u4"), ("row", "u4"), ("hwid", "u4"), ("xcc", "u4"), ("flags", "u4"),
                ("pad", "u4"), ("a", "f4", 4), ("b", "f4", 4), ("c", "f4", 4), ("t", "u8"), ("addr", "u8")])
CAP = 4096
log = torch.zeros(CAP * REC.itemsize, dtype=torch.uint8, device=dev)
nlog = torch.zeros(1, dtype=torch.int32, device=dev)
launch_no = [0]


_mod = {}
TRACE_NAMES = ["out.x", "out.y", "out.z",
               "pk_mul src0.lo (v6 = p5)", "pk_mul src0.hi (v7 = p0)", "pk_mul src1.lo (v2 = cx)",
               "pk_mul src1.hi (v3 = cy)", "pk_mul result.lo (v6)", "pk_mul result.hi (v7)",
               "pk_fma#1 result.lo (v0)", "pk_fma#1 result.hi (v1)", "pk_fma#2 result.lo (v0)",
               "pk_fma#2 src0.lo (v6 = p6)", "pk_fma#2 src0.hi (v7 = p2)", "pk_fma#2 src1 (v4 = cz)",
               "pk_fma#2 result.hi (v1)", "pk_add src0.lo (v18 = p3)", "pk_add src0.hi (v19 = p7)"]


def asm_victim(name, a):
    """The pointmap kernel from a hand-edited copy of its own assembly (tools/probes/race_asm_variants.sh -> pointmap_<name>.co),
    loaded through the HIP module API."""
    if not _mod:
        hip = C.CDLL("libamdhip64.so")
        m, f = C.c_void_p(), C.c_void_p()
        path = os.path.join(ROOT, "tools", "probes", f"pointmap_{name}.co").encode()
        assert hip.hipModuleLoad(C.byref(m), path) == 0, "hipModuleLoad " + path.decode()
        assert hip.hipModuleGetFunction(C.byref(f), m, b"_Z15pointmap_kernelPKfS0_S0_Pfiiiiji") == 0
        hip.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 7 + [C.c_void_p, C.c_void_p, C.c_void_p]
        _mod.update(hip=hip, f=f, m=m)
    vals = [C.c_void_p(_p(depth)), C.c_void_p(_p(K)), C.c_void_p(p_ptr), C.c_void_p(_p(a)), C.c_int(F), C.c_int(H), C.c_int(W),
            C.c_int(ratio), C.c_uint(seed), C.c_int(spf)]
    params = (C.c_void_p * len(vals))(*[C.cast(C.pointer(v), C.c_void_p) for v in vals])
    rc = _mod["hip"].hipModuleLaunchKernel(_mod["f"], (n + 255) // 256, 1, 1, 256, 1, 1, 0, _stream(), params, None)
    assert rc == 0, rc


vlib.race_aggr2_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
vlib.race_aggr_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
vlib.race_canary_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
NR = 96
SENT = torch.tensor([0x5A00005A | (r << 8) for r in range(NR)], dtype=torch.int64).to(torch.int32)


def canary():
    d = torch.empty(NR, n, dtype=torch.int32, device=dev)
    rc = vlib.race_canary_launch(_stream(), _p(d), n, args.iters)
    assert rc == 0, rc
    return d


def victim():
    if args.victim == "canary":
        return canary()
    a = torch.empty(n, 3 if args.victim in ("real", "full", "cnt", "drain", "nops") or args.victim.startswith("asm:") else 1, device=dev)
    if args.victim == "asm:trace":  # plane 0 = the output, planes 1-5 = registers captured along the pose product (see TRACE_NAMES)
        a = torch.full((6, n, 3), float("nan"), device=dev)
    if args.victim.startswith("asm:"):
        asm_victim(args.victim[4:], a)
    elif args.victim == "real":
        _lib.check(lib.l4p_point_map_samples(_stream(), _p(depth), _p(K), p_ptr, _p(a), F, H, W, ratio, seed), "pointmap")
    elif args.victim in ("full", "drain", "nops"):
        mode = {"full": 0, "drain": 1, "nops": 2}[args.victim] << 30
        rc = vlib.race_victim_full_launch(_stream(), _p(depth), _p(K), p_ptr, _p(a), F, H, W, ratio, seed, launch_no[0] | mode, _p(log),
                                          _p(nlog), CAP)
        assert rc == 0, rc
    elif args.victim == "cnt":
        rc = vlib.race_victim_cnt_launch(_stream(), _p(depth), _p(K), p_ptr, _p(a), F, H, W, ratio, seed, args.iters, launch_no[0],
                                         _p(log), _p(nlog), CAP)
        assert rc == 0, rc
    else:
        fn = vlib.race_victim_launch if args.victim == "plain" else vlib.race_victim_seq_launch
        rc = fn(_stream(), p_ptr, F, spf, args.iters, launch_no[0], _p(log), _p(nlog), CAP, _p(a))
        assert rc == 0, rc
    launch_no[0] += 1
    return a


cfg = ModelCfg.mini()
model = build(cfg, seeded_state_dict(cfg), "bf16")
net = model.l4p_model
batch = make_batch(args.frames, args.queries)
data = {k: (v.to(net.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
nwin = len(net.time_strides(args.frames))
with torch.no_grad():
    groups = parallel.encode_local_windows(net, data, ["track_2d"], 0, 1, 8)
    lasts = parallel.local_last_features(groups, 1)
    wins = [parallel.DecodedWindow(net.cfg.depth, {}, lasts[w]["last"]) for w in range(nwin)]
    trk = net.task_heads["track_2d"]
    strides = net.time_strides(args.frames)
    ref = victim().clone()
    torch.cuda.synchronize()
    nlog.zero_()
    bad = runs = 0
    first_bad = []
    side = torch.cuda.Stream()
    sink = torch.zeros(256, device=dev)
    srcbuf = torch.ones(4096, device=dev)
    for rep in range(args.reps):
        keep = None
        if args.aggressor == "tracker":
            trk.own_stream = trk.defer_join = True  # the recursion on a stream of its own, joined below
            keep = trk.forward_windowed(enc_features_bpc_2dlist=wins, time_strides=strides, **data)
        if args.aggressor.startswith("mfma:"):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(args.aggr_launches):
                    rc = vlib.race_aggr2_launch(_stream(), int(args.aggressor[5:]), args.aggr_blocks, args.aggr_iters, _p(sink))
                    assert rc == 0, rc
        if args.aggressor.startswith("synth:"):
            kind = int(args.aggressor[6:])
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(args.aggr_launches):
                    rc = vlib.race_aggr_launch(_stream(), kind, args.aggr_blocks, args.aggr_iters, _p(sink), _p(srcbuf))
                    assert rc == 0, rc
        res = [victim() for _ in range(args.per_rep)]
        if args.aggressor.startswith("synth:") or args.aggressor.startswith("mfma:"):
            torch.cuda.current_stream().wait_stream(side)
        if args.aggressor == "tracker":
            trk.join_streams()
            trk.own_stream = trk.defer_join = False
        torch.cuda.synchronize()
        for j, r in enumerate(res):
            runs += 1
            if not torch.equal(r, ref):
                bad += 1
                if args.victim == "canary":
                    want = SENT.to(dev)[:, None].expand(NR, n)
                    ij = (r != want).nonzero()
                    regs = sorted(set(ij[:, 0].tolist()))
                    thr = ij[:, 1]
                    print(f"   rep {rep} launch {j}: {ij.shape[0]} register values changed; registers {regs[:24]}; lanes "
                          f"{sorted(set((thr % 64).tolist()))[:20]}...; waves {sorted(set((thr // 64).tolist()))[:8]}; values "
                          f"{[hex(v & 0xFFFFFFFF) for v in r[ij[:6, 0], ij[:6, 1]].tolist()]}")
                    continue
                if args.victim == "asm:trace" and bad <= 4:
                    pts = (r[0] != ref[0]).any(dim=1).nonzero().flatten()
                    print(f"   rep {rep} launch {j}: {pts.numel()} points differ, first {int(pts[0])} (wave {int(pts[0]) // 64} lane {int(pts[0]) % 64})")
                    for pt in pts[:2].tolist() + pts[-1:].tolist():
                        got = r[:, pt].flatten().tolist()
                        want = ref[:, pt].flatten().tolist()
                        for nm, gv, wv in zip(TRACE_NAMES, got, want):
                            if gv != wv:
                                print(f"      point {pt}: {nm}: got {gv!r} want {wv!r}")
                if len(first_bad) < 6:
                    idx = (r != ref).flatten().nonzero().flatten()
                    first_bad.append((rep, j, int(idx[0]), int(idx[-1]), idx.numel(), r.flatten()[idx[:3]].tolist(),
                                      ref.flatten()[idx[:3]].tolist()))
        del keep, res
print(f"victim={args.victim} iters={args.iters} table={args.table} aggressor={args.aggressor}: "
      f"launches whose OUTPUT differs from the reference launch: {bad} / {runs}")
for fb in first_bad:
    print("   rep %d launch %d: flat %d..%d, %d values, got %s want %s" % fb)
if args.victim == "canary":
    sys.exit(0)
if args.victim != "real" and not args.victim.startswith("asm:"):
    cnt = int(nlog.item())
    print(f"in-kernel mismatches logged: {cnt} (lane-level records, cap {CAP})")
    recs = np.frombuffer(log.cpu().numpy().tobytes(), dtype=REC)[:min(cnt, CAP)]
    if cnt:
        lanes = recs["gid"] % 64
        print("   lanes:", dict(zip(*np.unique(lanes // 16 * 16, return_counts=True))), "(first lane of the quarter: count)")
        print("   rows / flags (1 = first load, 2 = second load, 4 = sc0 sc1 load, 8 = seq victim; row = bitmask for seq):",
              dict(zip(*np.unique(recs["row"] * 16 + recs["flags"], return_counts=True))))
        print("   iterations:", dict(zip(*np.unique(recs["iter"], return_counts=True))))
        print("   XCC:", dict(zip(*np.unique(recs["xcc"] & 0xF, return_counts=True))))
        hw = recs["hwid"]
        print("   CU id / SH / SE (HW_ID bits 8-11 / 12 / 13-15):", dict(zip(*np.unique(((hw >> 8) & 0xF) + 100 * ((hw >> 13) & 7),
                                                                                        return_counts=True))))
        print("   distinct (launch, wave):", len(set(zip(recs["launch"].tolist(), (recs["gid"] // 64).tolist()))))
        for r in recs[:8]:
            print("   rec launch %d iter %d gid %d (wave %d lane %d) row %d flags %d a=%s b=%s c=%s addr=%x" % (
                r["launch"], r["iter"], r["gid"], r["gid"] // 64, r["gid"] % 64, r["row"], r["flags"], r["a"].tolist(),
                r["b"].tolist(), r["c"].tolist(), r["addr"]))
