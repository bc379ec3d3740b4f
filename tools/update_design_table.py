"""Rewrite the round's measurement table in DESIGN.md (between the R4TABLE markers) from profiles/<tag>_*_bench_line.json.
usage: python tools/update_design_table.py [tag=r04]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(ROOT, "profiles")


def L(n):
    return json.load(open(os.path.join(P, f"{tag}_{n}_bench_line.json")))


c3, c2, c5, dm, pr = L("c3"), L("c2"), L("c5"), L("demo"), L("prep")


def kc(d, k, tf=True):
    v = d["kernel_classes"].get(k)
    if not v:
        return "–"
    s = f"{v['ms_per_step']:.1f} ms"
    if tf and v.get("tflops"):
        s += f" @ {v['tflops']:.0f} TF/s ({v['tflops'] / 25:.0f} %)"
    return s


rows = [("c3: all heads, B=4, 64 queries/clip", c3), ("c2: depth only, B=1", c2),
        ("c5: one 256-frame video (31 windows), all heads, 1 GPU, windows in groups of 16", c5),
        ("demo: 64 frames, 625 queries in chunks of 128, depth + flow + mask + tracks", dm)]
b8 = os.path.join(P, f"{tag}_c3_batch8_bench_line.json")
if os.path.exists(b8):
    rows.insert(1, ("c3 at batch 8 (the per-GPU batch of configs[3])", json.load(open(b8))))
t = ("<!--R4TABLE-BEGIN-->\n| workload | frames/s | ms/step | GEMM class | conv3d class | attention | small / streaming products | LayerNorm | elementwise | tracker kernels |\n"
     "|---|---|---|---|---|---|---|---|---|---|\n")
for name, d in rows:
    t += (f"| {name} | **{d['value']:.0f}** | {d['ms_per_step']:.1f} | {kc(d, 'gemm')} | {kc(d, 'conv3d')} | {kc(d, 'attention')} | {kc(d, 'gemm_small')} | "
          f"{kc(d, 'layernorm', False)} | {kc(d, 'elementwise', False)} | {kc(d, 'track', False)} |\n")
t += f"| prep: 50 decoded 480×854 frames → [3,64,224,224] | **{pr['value'] / 1000:.0f} k** | {pr['ms_per_step']:.2f} | | | | | | | |\n"
attn = [l for l in open(os.path.join(P, f"{tag}_c3_kernel_stats.md")) if "attn64_kernel" in l or "_Z11attn_kernel" in l][0].split("|")
avg = float(attn[4])
rd = c3["roofline"]
ra = c3.get("roofline_attention", rd)
rg = c3.get("roofline_gemm", rd)
rc = c3.get("roofline_conv3d", rd)
t += (f"\nThe c3 line: `roofline` = the class with the most time, {rd['kernel'].split(' ')[0]}: {rd['achieved']:.0f} TF/s = **{rd['frac']:.3f}** of 2.5 PF; "
      f"GEMM class (`roofline_gemm`) {rg['achieved']:.0f} TF/s = **{rg['frac']:.3f}** ({rg['launches_per_step']:.0f} launches per step, HIP-event-timed); "
      f"conv3d class {rc['achieved']:.0f} TF/s = **{rc['frac']:.3f}**; `roofline_attention` {ra['achieved']:.0f} TF/s = **{ra['frac']:.3f}** "
      f"({ra['avg_launch_us']:.1f} µs per launch event-timed; rocprofv3 average of the same command {avg:.2f} µs = {94.49 / avg / 2.5:.3f}, "
      f"`profiles/{tag}_c3_kernel_stats.md`; c5: {c5['roofline_attention']['frac']:.3f} at its batch of 16 windows).  HBM bytes per launch (PMC): "
      f"{(rd.get('traffic') or 0) / 1e6:.0f} MB ({rd['kernel'].split(' ')[0]}), {(rg.get('traffic') or 0) / 1e6:.0f} MB (GEMM class), {(ra.get('traffic') or 0) / 1e6:.0f} MB (attention; `profiles/{tag}_c3_hbm_traffic.md`).  CPU oracle on the same box: {c3['cpu_baseline']['value']:.2f} frames/s on "
      f"{c3['cpu_baseline']['cores']} cores ({c3['cpu_baseline']['sample'].split(';')[1].strip()}).\n")
g = c5.get
t += (f"c5 pieces on one GPU: encoders {g('phase1a_encoder_ms'):.0f} ms, decoders {g('phase1b_decoders_ms'):.0f} ms, dense stitch {g('phase3_dense_ms'):.1f} ms, "
      f"tracker {g('phase3_track_ms'):.0f} ms ({g('phase3_track_ms_on_an_eighth_of_the_queries'):.0f} ms on an eighth of the queries), exchanges on one rank "
      f"{g('exchange_last_ms'):.1f} + {g('exchange_decoded_ms'):.1f} ms.  **One of eight ranks, emulated and measured: {g('emulated_rank0_of_8_ms'):.0f} ms "
      f"= {g('implied_8gpu_speedup_emulated_rank'):.1f}x** implied on 8 GPUs with the gather schedule"
      + (f", **{g('emulated_rank0_of_8_seam_local_ms'):.0f} ms = {g('implied_8gpu_speedup_emulated_rank_seam_local'):.1f}x with the seam-local exchange** "
         f"(the rank's dense path {g('phase3_dense_seam_local_rank0_of_8_ms'):.1f} ms instead of {g('phase3_dense_ms'):.1f}; bytes received per rank for the dense path "
         f"{g('exchange_bytes_per_rank_of_8')['dense_seam_local_schedule'] / 1e6:.1f} MB instead of {g('exchange_bytes_per_rank_of_8')['dense_gather_schedule'] / 1e6:.0f} MB, "
         f"+ {g('exchange_bytes_per_rank_of_8')['last_layer_features_all_gather'] / 1e6:.0f} MB of last-layer features for the tracker in both)"
         if g('emulated_rank0_of_8_seam_local_ms') else "")
      + f" (models: tracker after the decoders {g('implied_8gpu_speedup_tracker_after_decoders'):.1f}x, ideally beside them {g('implied_8gpu_speedup_tracker_beside_decoders'):.1f}x).\n<!--R4TABLE-END-->\n")
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
i, j = s.index("<!--R4TABLE-BEGIN-->"), s.index("<!--R4TABLE-END-->") + len("<!--R4TABLE-END-->\n")
open(path, "w").write(s[:i] + t + s[j:])
print(t)
