#!/bin/bash
# what the driver runs at round end, in the same order: GPU tests, smoke(), the default bench line (+ the N = 2 plumbing check)
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_final.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest_final.log
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | cut -c1-300
timeout 900 python bench.py --steps 50 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_final.json').read().strip().splitlines()[-1])
r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','steps','warmup')}, d['eval']['value'], d['parity']['ok'], r['frac'], r['traffic'], d['cpu_baseline']['value'])
PY
