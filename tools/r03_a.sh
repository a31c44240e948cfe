#!/bin/bash
# round 3, session A: the GPU parity suite on the new tests + the default bench line (baseline of this box)
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest_a.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_a.log
tail -15 $OUT/pytest_a.log
timeout 600 python bench.py > $OUT/bench_a.json 2> $OUT/bench_a.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_a.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_hip_events','n_ranks_seen')}, d['eval']['value'], d['parity']['ok'], d['parity']['embeddings_after_steps_max_rel'], d['parity']['param_max_rel'])
print({k:d['roofline'][k] for k in ('frac','ms_per_step','traffic')}, d['roofline']['second']['frac'], d['roofline']['second']['ms_per_step'])
print(d['row_sharded']['strong'].get('ms_per_step'), d['row_sharded']['strong'].get('config',{}).get('edges_per_gpu'), d['row_sharded']['weak'].get('ms_per_step'))
PY
