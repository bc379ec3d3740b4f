"""Report for tools/probes/c5_rank_timeline.py: python c5_rank_timeline_report.py <dir with *kernel_trace.csv>
Cuts the kernel trace at idle gaps > 0.3 s, keeps the last three segments (A, B, C) and prints per segment: span, number of
kernels, busy time (union of kernel intervals) per queue, the time during which kernels of two different queues ran at once, the
idle time (no kernel running), and the largest kernels per queue."""
import csv
import glob
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def overlap(a, b):
    """time during which at least one interval of a AND one of b are open"""
    ev = [(s, 0, 1) for s, _ in a] + [(e, 0, -1) for _, e in a] + [(s, 1, 1) for s, _ in b] + [(e, 1, -1) for _, e in b]
    ev.sort(key=lambda x: (x[0], x[2]))
    n = [0, 0]
    last, tot = None, 0
    for t, w, d in ev:
        if last is not None and n[0] > 0 and n[1] > 0:
            tot += t - last
        n[w] += d
        last = t
    return tot


def main():
    d = sys.argv[1]
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    segs, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - max(x[1] for x in cur[-50:]) > 300_000_000:
            segs.append(cur)
            cur = []
        cur.append(r)
    segs.append(cur)
    print(f"{len(rows)} kernels, {len(segs)} segments; reporting the last 3")
    for name, seg in zip("ABC", segs[-3:]):
        s0, e0 = min(r[0] for r in seg), max(r[1] for r in seg)
        byq = defaultdict(list)
        for s, e, q, k in seg:
            byq[q].append((s, e))
        print(f"\nsegment {name}: span {(e0 - s0) / 1e6:.2f} ms, {len(seg)} kernels, busy (any queue) {union([(s, e) for s, e, _, _ in seg]) / 1e6:.2f} ms")
        qs = sorted(byq, key=lambda q: -union(byq[q]))
        for q in qs:
            iv = byq[q]
            print(f"  queue {q}: {len(iv)} kernels, first start +{(min(s for s, _ in iv) - s0) / 1e6:.2f} ms, last end +{(max(e for _, e in iv) - s0) / 1e6:.2f} ms, "
                  f"busy {union(iv) / 1e6:.2f} ms, sum of durations {sum(e - s for s, e in iv) / 1e6:.2f} ms")
        for i in range(len(qs)):
            for j in range(i + 1, len(qs)):
                o = overlap(byq[qs[i]], byq[qs[j]])
                if o:
                    print(f"  queues {qs[i]} and {qs[j]} both running: {o / 1e6:.2f} ms")
        if name == "A":
            for q in qs:
                ks = sorted((s, e, k) for s, e, qq, k in seg if qq == q)
                print(f"  queue {q} first kernels: " + "; ".join(f"+{(s - s0) / 1e6:.2f} {k[:28]}" for s, e, k in ks[:4]))
                conv = [s for s, e, k in ks if "conv3" in k or "upsample" in k]
                if conv:
                    print(f"  queue {q}: first decoder kernel (conv / upsample) at +{(min(conv) - s0) / 1e6:.2f} ms")
        # duration of the tracker-side kernels by name within this segment (top 12 by total)
        agg = defaultdict(lambda: [0, 0])
        for s, e, q, k in seg:
            a = agg[(q, k[:70])]
            a[0] += e - s
            a[1] += 1
        top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]
        for (q, k), (t, n) in top:
            print(f"    q{q} {t / 1e6:7.2f} ms {n:5d} x {t / n / 1e3:7.1f} us  {k}")


if __name__ == "__main__":
    main()
