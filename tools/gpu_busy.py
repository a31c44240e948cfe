"""GPU busy fraction of the steps of a Python-driven (not graph-replayed) loop in a rocprofv3 kernel trace: steps are delimited by a
kernel that is launched once per step (default: the first name containing `sample`); for the steps of the second half of the run the
union of all kernel intervals is compared with the step's span - the idle remainder is what HIP-graph capture could remove.
python tools/gpu_busy.py <results.db> [delimiter substring]"""
import sqlite3
import sys

db = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "sample"
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name}, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if key in r[0]]
if len(marks) < 4:
    from collections import Counter
    print("delimiter %r found %d times; kernel names:" % (key, len(marks)), Counter(r[0].split("(")[0][-50:] for r in rows).most_common(30))
    sys.exit(1)
steps = [(marks[i], marks[i + 1]) for i in range(len(marks) // 2, len(marks) - 1)]
tot_span = tot_busy = n_k = 0
gaps = []
for lo, hi in steps:
    seg = rows[lo:hi]
    cur_s, cur_e, busy = seg[0][1], seg[0][2], 0
    for _, s, e in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = rows[hi][1] - seg[0][1]
    gaps.append(rows[hi][1] - cur_e)
    tot_span += span; tot_busy += busy; n_k += len(seg)
gaps.sort()
print("steps %d  kernels/step %.1f  step %.3f ms  GPU busy %.3f ms (%.1f %%)  idle/step %.1f us in %d gaps (median %.1f us, largest %.1f us)" % (
    len(steps), n_k / len(steps), tot_span / len(steps) / 1e6, tot_busy / len(steps) / 1e6, 100.0 * tot_busy / tot_span,
    (tot_span - tot_busy) / len(steps) / 1e3, len(gaps) // len(steps), gaps[len(gaps) // 2] / 1e3, gaps[-1] / 1e3))
