"""One steady-state training step as a kernel timeline (from a rocprofv3 rocpd database of bench.py).
python tools/step_timeline.py <results.db> <out.txt>  - picks a steady-state step (delimited by the in-graph sampler launch)."""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
sel = f"select {name}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
rows = c.execute(sel).fetchall()
# a step = the launches from one in-graph sampler launch up to the next one; inside a graph of several steps the projection and the
# sampler are both first (either may start a few us ahead of the other): a projection launched within 40 us before the sampler
# belongs to the sampler's step
starts = [i for i, r in enumerate(rows) if "sample_batch_kernel" in r[0]]
for k, i in enumerate(starts):
    j = i - 1
    while j >= 0 and rows[i][1] - rows[j][1] < 40000:
        if "linear_fwd_grouped" in rows[j][0]:
            starts[k] = j
            break
        j -= 1
cand = [(starts[i - 1], starts[i]) for i in range(1, len(starts))]
lo, hi = cand[len(cand) // 2]                                  # a steady-state step from the middle of the run
step = rows[lo: hi]
t0 = step[0][1]
with open(out, "w") as f:
    f.write("# start_us  dur_us  gap_since_prev_end_us  stream  kernel\n")
    prev_end = t0
    for n, s, e, q in step:
        f.write("%9.2f %8.2f %8.2f  %s  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, n.split("(")[0][-70:]))
        prev_end = max(prev_end, e)
    f.write("# step span %.2f us (first launch to the last end), %.2f us to the next step's first launch, kernel time sum %.2f us, %d kernels\n" % (
        (max(e for _, s, e, _ in step) - t0) / 1e3, (rows[hi][1] - t0) / 1e3, sum(e - s for _, s, e, _ in step) / 1e3, len(step)))
print(open(out).read()[-200:])
