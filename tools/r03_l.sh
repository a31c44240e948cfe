#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_bench_shapes.py tests/test_gpu_options.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "bpr or adamw or step or netflix or ml_shape or golden or options or fused" > $OUT/pytest_l.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_l.log
tail -8 $OUT/pytest_l.log
timeout 900 python bench.py --no-cpu-baseline --no-row-sharded > $OUT/bench_l.json 2> $OUT/bench_l.err; echo "bench exit $?"; tail -3 $OUT/bench_l.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_l.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_hip_events')}, d['eval']['value'], d['parity']['ok'], d['parity']['embeddings_after_steps_max_rel'])
r=d['roofline']; print({k:r[k] for k in ('frac','ms_per_step','ms_isolated','traffic','timing')}); print({k:r['second'][k] for k in ('frac','ms_per_step','ms_isolated','traffic')})
print(d.get('exact_f32',{}).get('ms_per_step'), d.get('reference_order'))
print([k.get('insitu_error') for k in d['kernels']])
PY
rm -rf /tmp/prof_l
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o bench -- python $OLDPWD/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded > $OLDPWD/$OUT/prof_l.log 2>&1; echo "prof exit $?")
DB=$(find /tmp/prof_l -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $OUT/bench_nf_kernel_stats_l.csv 121 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded (121 steps + 7 evaluations)"
python tools/step_timeline.py $DB $OUT/step_timeline_l.txt > /dev/null
cat $OUT/step_timeline_l.txt
