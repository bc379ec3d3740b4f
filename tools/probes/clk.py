"""Effective shader clock per kernel from a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace run: python clk.py <dir>"""
import sqlite3, sys, collections
d = sys.argv[1]
db = sqlite3.connect(d + "/out_results.db"); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
pmc = [t for t in tabs if "pmc_event" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
pc = [r[1] for r in cur.execute(f"pragma table_info({pmc})")]
ev = "event_id" if "event_id" in pc else pc[1]
rows = cur.execute(f"select k.kernel_id, k.end-k.start, (select sum(p.value) from {pmc} p where p.{ev}=k.event_id) from {kd} k").fetchall()
agg = collections.defaultdict(list)
for kid, dur, act in rows:
    if act:
        agg[names.get(kid, str(kid))[:60]].append((dur, act))
for n, v in agg.items():
    v = v[len(v) // 4:]
    dur = sum(x[0] for x in v) / len(v); act = sum(x[1] for x in v) / len(v)
    print(f"{n:60s} n={len(v):3d} dur {dur/1e3:8.1f} us  GUI_ACTIVE {act:10.0f}  clock {act/dur:5.2f} GHz")
