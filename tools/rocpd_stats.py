"""Per-kernel summary (CSV) from a rocprofv3 rocpd database: python tools/rocpd_stats.py <results.db> <out.csv> <steps> "<header comment>" """
import sqlite3, sys
db, out, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
note = sys.argv[4] if len(sys.argv) > 4 else ""
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
start = "start" if "start" in cols else "start_time"
end = "end" if "end" in cols else "end_time"
rows = c.execute(f"select {name}, count(*), sum({end}-{start})/1e3, avg({end}-{start})/1e3 from kernels group by {name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    if note:
        f.write("# %s\n" % note)
    f.write("# durations in microseconds; per_step_us = total_us / %g\n" % steps)
    f.write("name,calls,total_us,avg_us,pct,per_step_us\n")
    for n, k, t, a in rows:
        f.write('"%s",%d,%.3f,%.3f,%.2f,%.2f\n' % (n, k, t, a, 100 * t / tot, t / steps))
print("kernels", len(rows), "total us", tot)
