#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_t
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o rs -- python $OLDPWD/bench.py --workload cfg4 --synth-scaling strong --steps 6 --warmup 2 --no-parity --no-kernel-roofline > $OLDPWD/$OUT/prof_t.log 2>&1; echo "prof exit $?")
DB=$(find /tmp/prof_t -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name}, start, end from kernels order by start").fetchall()
# last step: print every spmm launch's duration in order
idx = [i for i, r in enumerate(rows) if "sample" in r[0]]
lo = idx[-2]; hi = idx[-1]
for n, s, e in rows[lo:hi]:
    if (e - s) > 200000: print("%8.1f us  %s" % ((e - s) / 1e3, n.split("(")[0][-60:]))
PY
