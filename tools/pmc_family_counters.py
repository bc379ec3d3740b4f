"""Per-kernel-family averages of arbitrary PMC counters from several rocprofv3 passes (sqlite output, one directory per pass):

  python tools/pmc_family_counters.py <dir> [<dir> ...] > profiles/rNN_c3_wave_state_counters.md

Each pass: rocprofv3 --pmc <up to 4 SQ counters> --kernel-trace -d <dir> -o out -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof
(separate passes, --kernel-trace only).  Values are the chip-wide sums per launch, averaged over the launches of the family; the
ratios printed under the table divide by SQ_WAVE_CYCLES (wave-resident cycles) or by the instruction counts."""
import collections
import sys

from pmc_mfma_util import HOT, load


def main():
    tables = [load(d) for d in sys.argv[1:]]
    fam = collections.OrderedDict()
    for key, label in HOT:
        row = {}
        n = 0
        for src in tables:
            per = collections.defaultdict(list)
            for k, cs in src.items():
                if key.search(k):
                    for c, lst in cs.items():
                        per[c].extend(v for _, v in lst)
            for c, vals in per.items():
                row[c] = sum(vals) / len(vals)
                n = max(n, len(vals))
        if row:
            fam[label] = (n, row)
    counters = sorted({c for _, (_, r) in fam.items() for c in r})
    print("# Wave-state / instruction-mix counters per hot kernel family (PMC, c3 step)\n")
    print("`rocprofv3 --pmc <counters> --kernel-trace` passes of `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof`; chip-wide sums per launch, "
          "averaged over the family's launches.\n")
    print("| kernel family | launches | " + " | ".join(counters) + " |")
    print("|---|---|" + "---|" * len(counters))
    for label, (n, r) in fam.items():
        print(f"| {label} | {n} | " + " | ".join(f"{r[c]:.3g}" if c in r else "" for c in counters) + " |")
    print("\n| kernel family | waiting (any) / wave cycles | waiting for an instruction slot / wave cycles | issuing / wave cycles | LDS bank-conflict cycles / LDS-active cycles | LDS instr per MFMA | VALU (non-MFMA) instr per MFMA | SALU instr per MFMA | VMEM instr per MFMA |")
    print("|---|---|---|---|---|---|---|---|---|")
    for label, (n, r) in fam.items():
        g = lambda c: r.get(c)  # noqa: E731
        def ratio(a, b):
            return f"{g(a) / g(b):.3f}" if g(a) is not None and g(b) else ""
        mf = g("SQ_INSTS_MFMA") or g("SQ_INSTS_VALU_MFMA_MOPS_BF16")
        def per_mfma(a, minus=None):
            if g(a) is None or not mf:
                return ""
            v = g(a) - (g(minus) if minus and g(minus) else 0)
            return f"{v / mf:.2f}"
        print(f"| {label} | {ratio('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')} | {ratio('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES')} | {ratio('SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES')} | "
              f"{ratio('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')} | {per_mfma('SQ_INSTS_LDS')} | {per_mfma('SQ_INSTS_VALU', 'SQ_INSTS_MFMA')} | {per_mfma('SQ_INSTS_SALU')} | {per_mfma('SQ_INSTS_VMEM')} |")


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
