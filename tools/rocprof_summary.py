"""rocprofv3 (rocpd sqlite .db or *_kernel_trace.csv) -> per-kernel statistics table (markdown).
Usage: python tools/rocprof_summary.py <results.db|kernel_trace.csv> [title]"""
import csv
import sqlite3
import sys


def rows_from_db(path):
    c = sqlite3.connect(path)
    return c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                     "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()


def rows_from_csv(path):
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a = agg.setdefault(r["Kernel_Name"], [0, 0.0, 1e30, 0.0, 0, 0, 0])
            a[0] += 1
            a[1] += d
            a[2] = min(a[2], d)
            a[3] = max(a[3], d)
            a[4] = max(a[4], int(r.get("VGPR_Count", 0) or 0))
            a[5] = max(a[5], int(r.get("Accum_VGPR_Count", 0) or 0))
            a[6] = max(a[6], int(r.get("LDS_Block_Size", 0) or 0))
    out = [(k, v[0], v[1], v[1] / v[0], v[2], v[3], v[4], v[5], v[6]) for k, v in agg.items()]
    return sorted(out, key=lambda r: -r[2])


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary: {title}\n")
    print(f"total kernel time {tot/1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total us | avg us | min us | max us | % | vgpr | agpr | lds |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| `{r[0][:110]}` | {r[1]} | {r[2]:.1f} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | {100*r[2]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} |")


if __name__ == "__main__":
    main()
