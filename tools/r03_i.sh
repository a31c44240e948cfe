#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest_i.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_i.log
tail -6 $OUT/pytest_i.log
timeout 600 python bench.py --no-cpu-baseline --no-row-sharded > $OUT/bench_i.json 2> $OUT/bench_i.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_i.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_hip_events')}, d['eval']['value'], d['parity']['ok'], d['parity']['embeddings_after_steps_max_rel'])
print({k:d['roofline'][k] for k in ('frac','ms_per_step','traffic')}, d['roofline']['second']['frac'], d['roofline']['second']['ms_per_step'], d.get('exact_f32',{}).get('ms_per_step'))
PY
rm -rf /tmp/prof_i
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o bench -- python $OLDPWD/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded > $OLDPWD/$OUT/prof_i.log 2>&1; echo "prof exit $?")
DB=$(find /tmp/prof_i -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $OUT/bench_nf_kernel_stats_i.csv 121 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded (121 steps + 7 evaluations)"
python tools/step_timeline.py $DB $OUT/step_timeline_i.txt > /dev/null
cat $OUT/step_timeline_i.txt
