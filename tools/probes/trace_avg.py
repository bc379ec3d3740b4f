"""Average kernel duration from a rocprofv3 --kernel-trace results db: python trace_avg.py <dir>..."""
import sqlite3, sys
for d in sys.argv[1:]:
    db = sqlite3.connect(d + "/out_results.db"); cur = db.cursor()
    kd = [r[0] for r in cur.execute("select name from sqlite_master where type='table'") if "kernel_dispatch" in r[0]][0]
    rows = cur.execute(f"select end-start from {kd} order by start").fetchall()
    v = sorted(r[0] for r in rows[5:])
    print(d.split('/')[-1], f"n={len(v)} median {v[len(v)//2]/1e3:.1f} us  mean {sum(v)/len(v)/1e3:.1f} us  min {v[0]/1e3:.1f}")
