"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output) -> markdown + json.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof
  python tools/pmc_hbm_traffic.py out/fetch out/write profiles/r02_c3_hbm_traffic

Units (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count kilobytes... on gfx950 FETCH_SIZE
under-reports by 2x (checked in-run on a kernel whose input size is known).  Kernel classes as in csrc/prof.hpp."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = r["Kernel_Name"]
                per[k][0] += 1
                per[k][1] += float(r["Counter_Value"])
    return per


def klass(name):
    if "gemm8p_kernel<1" in name or "conv3_halo" in name or ("gemm_kernel" in name and "ELi1ELb" in name):
        return "conv3d"
    # (kernel names carry no shapes: the 128-row-tile kernel is counted with the small / streaming products it mostly serves;
    #  bench.py's event profiler classes launches by rows / row-grouped weights, csrc/prof.hpp)
    if "t2i_ctx" in name or "i2t_delta" in name or "gemm_group" in name or "gemm_kernel" in name:
        return "gemm_small"
    if "gemm" in name or "splitk" in name:
        return "gemm"
    if ("attn_kernel" in name or "attn64_kernel" in name) and "t2i" not in name and "i2t" not in name and "self" not in name:
        return "attention"
    if "layernorm" in name:
        return "layernorm"
    if any(t in name for t in ("track", "t2i", "i2t", "self_attn6", "mask_", "fill_rows")):
        return "track"
    if any(t in name for t in ("upsample", "head_out", "cast_kernel", "patch_gather", "rays_to", "affine", "umeyama")):
        return "elementwise"
    return None


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    rows = []
    cls = defaultdict(lambda: [0, 0.0, 0.0])
    for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
        n = max(fe.get(k, [0, 0])[0], wr.get(k, [0, 0])[0])
        rd = 2.0 * fe.get(k, [0, 0.0])[1] * 1024 / max(n, 1)   # KB -> B, x2 (gfx950 correction)
        wt = wr.get(k, [0, 0.0])[1] * 1024 / max(n, 1)
        rows.append((k, n, rd, wt))
        c = klass(k)
        if c:
            cls[c][0] += n
            cls[c][1] += rd * n
            cls[c][2] += wt * n
    with open(out + ".md", "w") as f:
        f.write(f"# HBM traffic per launch (PMC), {os.path.basename(out)}, final tree\n\n")
        f.write("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) of `python bench.py --steps 1 "
                "--warmup 1 --no-cpu-baseline --no-prof` (c3, 2 steps); FETCH_SIZE doubled per the gfx950 correction of "
                "MI355X_MICROARCH.md, WRITE_SIZE as reported.\n\n| kernel | launches | HBM read MB/launch | HBM write MB/launch |\n|---|---|---|---|\n")
        for k, n, rd, wt in rows[:40]:
            f.write(f"| `{k[:100]}` | {n} | {rd / 1e6:.1f} | {wt / 1e6:.1f} |\n")
        f.write("\n| class | launches | HBM read MB/launch | HBM write MB/launch |\n|---|---|---|---|\n")
        for c, (n, rd, wt) in cls.items():
            f.write(f"| {c} | {n} | {rd / n / 1e6:.1f} | {wt / n / 1e6:.1f} |\n")
    js = {"per_class_bytes_per_launch": {c: {"launches": n, "hbm_read": rd / n, "hbm_write": wt / n, "hbm_total": (rd + wt) / n}
                                         for c, (n, rd, wt) in cls.items()},
          "method": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE, separate passes, bench.py c3 2 steps",
          "kernel_tree": _kernel_tree()}  # bench.py attaches these figures only to the kernels they were measured on
    with open(out + ".json", "w") as f:
        json.dump(js, f, indent=1)
    print(open(out + ".md").read()[-1200:])


def _kernel_tree():
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from l4p_amd._lib import kernel_tree_hash

    return kernel_tree_hash()


if __name__ == "__main__":
    main()
