"""FETCH_SIZE / WRITE_SIZE as reported by rocprofv3 against the bytes tools/probes/fetch_calib actually moved (1 GiB per launch).
  python tools/pmc_fetch_calibration.py <fetch csv dir> <write csv dir> > profiles/rNN_fetch_size_calibration.md"""
import csv
import glob
import os
import sys
from collections import defaultdict

BYTES = float(1 << 30)


def load(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("# FETCH_SIZE / WRITE_SIZE calibration on gfx950 (tools/probes/fetch_calib: 1 GiB streamed per launch, fully coalesced)\n")
print("| kernel | counter | reported KB (mean of launches) | reported bytes / moved bytes | factor to apply |\n|---|---|---|---|---|")
for name, acc, cnt in [(k, fe, "FETCH_SIZE") for k in sorted(fe) if "read_" in k] + [(k, wr, "WRITE_SIZE") for k in sorted(wr) if "write_w" in k]:
    v = sum(acc[name]) / len(acc[name])
    moved = BYTES if "rows8" not in name else (BYTES // 2816) * 2816
    ratio = v * 1024 / moved
    print(f"| `{name[:60]}` | {cnt} | {v:.0f} | {ratio:.3f} | x{1 / ratio:.2f} |")
