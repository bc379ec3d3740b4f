"""Matrix-pipe utilisation and effective clock per hot kernel from two rocprofv3 PMC passes (sqlite output):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d out/sq   -o out -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof
  rocprofv3 --pmc GRBM_GUI_ACTIVE                          --kernel-trace -d out/grbm -o out -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof
  python tools/pmc_mfma_util.py out/sq out/grbm [power.txt] > profiles/rNN_c3_mfma_util.md

Units (MI355X_MICROARCH.md, per-instruction cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles a SIMD's matrix pipe
is busy, summed over the chip (32 per v_mfma_f32_32x32x16_bf16, 16 per v_mfma_f32_16x16x32_bf16); GRBM_GUI_ACTIVE counts shader
cycles of the dispatch, so GUI_ACTIVE / wall = the effective clock and MFMA_BUSY / (GUI_ACTIVE x 1024 SIMDs) = the fraction
of the chip's MFMA issue cycles used AT THE CLOCK THE CHIP GRANTED (nominal-peak fraction = that x clock / 2.4 GHz)."""
import collections
import glob
import os
import sqlite3
import sys

SIMDS = 256 * 4
# (rocprofv3 stores some kernel names demangled and some mangled: the patterns are regular expressions over either spelling; the
#  engine-dtype template parameter - DF16b / DF16_ / f, "__bf16" / "_Float16" / "float" - comes first in every GEMM / conv family)
import re

_T = r"(?:DF16b|DF16_|f|__bf16, |_Float16, |float, )?"
HOT = ((rf"gemm8p_kernel(?:I{_T}Li0ELi2ELi4E|<{_T}0, 2, 4)", "gemm8p<0,2,4> (linear GEMMs, 256x256 tiles)"),
       (rf"gemm8p_kernel(?:I{_T}Li0ELi4ELi2E|<{_T}0, 4, 2)", "gemm8p<0,4,2> (linear GEMMs, 256x192 tiles)"),
       (rf"gemm8p_kernel(?:I{_T}Li0E|<{_T}0,)", "gemm8p<0> (both instantiations)"),
       (rf"gemm8p_kernel(?:I{_T}Li1E|<{_T}1,)", "gemm8p<1> (3x3x3 conv, implicit GEMM)"),
       (r"conv3_halo", "conv3_halo (3x3x3 conv, LDS halo)"),
       (rf"(?:11gemm_kernelI{_T}Li128ELi128ELi2ELi2ELi1|gemm_kernel<{_T}128, 128, 2, 2, 1)", "gemm_kernel 128x128 MODE1 (3x3x3 conv)"),
       (rf"(?:11gemm_kernelI{_T}Li128ELi128ELi2ELi2ELi0|gemm_kernel<{_T}128, 128, 2, 2, 0)", "gemm_kernel 128x128 MODE0"),
       (rf"(?:11gemm_kernelI{_T}Li128ELi64|gemm_kernel<{_T}128, 64)", "gemm_kernel 128x64 (token-side GEMMs)"),
       (r"attn64_kernel", "attn64_kernel (encoder attention, 64 query rows per wave)"),
       (r"(?:_Z11attn_kernelI\w+Li96E|void attn_kernel<\w+, 96)", "attn_kernel (encoder attention, 32 query rows per wave)"))
HOT = tuple((re.compile(k), label) for k, label in HOT)


def load(d):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    db = sqlite3.connect(dbs[0])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    pmc = [t for t in tabs if "pmc_event" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    pi = [t for t in tabs if "info_pmc" in t or "pmc_info" in t]
    names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
    pc = [r[1] for r in cur.execute(f"pragma table_info({pmc})")]
    ev = "event_id" if "event_id" in pc else pc[1]
    cname = {}
    if pi:
        cols = [r[1] for r in cur.execute(f"pragma table_info({pi[0]})")]
        nm = "name" if "name" in cols else "symbol"
        cname = {r[0]: r[1] for r in cur.execute(f"select id, {nm} from {pi[0]}")}
    pid = "pmc_id" if "pmc_id" in pc else None
    out = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> [(dur, value)]
    disp = {r[0]: (r[1], r[2]) for r in cur.execute(f"select event_id, kernel_id, end - start from {kd}")}
    q = f"select {ev}, {pid if pid else '0'}, sum(value) from {pmc} group by {ev}, {pid if pid else '0'}"
    for e, p, v in cur.execute(q):
        if e in disp:
            kid, dur = disp[e]
            out[names.get(kid, str(kid))][cname.get(p, str(p))].append((dur, v))
    return out


def main():
    sq, gr = load(sys.argv[1]), load(sys.argv[2])
    print("# Matrix-pipe utilisation and effective clock per hot kernel (PMC)\n")
    print("rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES` and `--pmc GRBM_GUI_ACTIVE` (two separate passes, `--kernel-trace` only) of "
          "`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof` (c3: batch 4, all heads; 2 steps).  Averages per launch over the "
          "launches of the kernel family; clock = GRBM_GUI_ACTIVE / wall; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs); "
          "'of nominal' = MFMA busy x clock / 2.4 GHz (the clock the 2.5 PF peak is quoted at).\n")
    print("| kernel family | launches | wall us (SQ pass) | wall us (GRBM pass) | GUI_ACTIVE cycles (per XCD) | clock GHz | MFMA busy cycles | MFMA busy % at the granted clock | % of nominal-clock MFMA cycles |")
    print("|---|---|---|---|---|---|---|---|---|")
    for key, label in HOT:
        def agg(src, counter):
            vals = [x for k, cs in src.items() if key.search(k) for c, lst in cs.items() if counter in c for x in lst]
            if not vals:
                return None
            return len(vals), sum(v[0] for v in vals) / len(vals), sum(v[1] for v in vals) / len(vals), sum(v[0] for v in vals), sum(v[1] for v in vals)
        m, g = agg(sq, "MFMA_BUSY"), agg(gr, "GUI_ACTIVE")
        if not m or not g:
            continue
        # time-weighted over the family: total busy / total active
        clock, gui = g[4] / g[3], g[2]
        if clock > 4.0:  # GRBM_GUI_ACTIVE reported as the SUM over the 8 XCDs' GRBM instances
            clock, gui = clock / 8.0, gui / 8.0
        busy = m[4] / (m[3] * clock * SIMDS)  # busy cycles / (wall of the SQ pass x clock x SIMDs)
        print(f"| {label} | {m[0]} | {m[1] / 1e3:.1f} | {g[1] / 1e3:.1f} | {gui:.0f} | {clock:.2f} | {m[2]:.0f} | {100 * busy:.1f} | {100 * busy * clock / 2.4:.1f} |")
    seen = sorted(((sum(x[0] for c, lst in cs.items() if "MFMA_BUSY" in c for x in lst), k) for k, cs in sq.items()), reverse=True)[:12]
    print("\n<!-- kernels by wall time in the SQ pass: " + "; ".join(f"{k[:70]} {t / 1e3:.0f}us" for t, k in seen) + " -->")
    print("\nEvery family that fills the chip runs at 1.9-2.1 GHz instead of 2.4 GHz: the board holds its power limit by lowering the clock "
          "under matrix-pipe load, so the last column (what the roofline fraction measures) is bounded by busy% x clock/2.4.")
    if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
        print("\n## Socket power and clock during the c3 step (`tools/probes/power_probe.sh`: rocm-smi sampled every 2 s over 400 steps; count, power W, sclk)\n")
        print("```")
        print(open(sys.argv[3]).read().strip())
        print("```")


if __name__ == "__main__":
    main()
