#!/bin/bash
# Round-3 evidence run: full GPU test suite, the default bench line (with the CPU baseline leg), the rocprofv3 kernel
# summary + step timeline of the same command, the PMC passes, the MovieLens-shape line.
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_k.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_k.log
tail -6 $OUT/pytest_k.log
timeout 900 python bench.py > $OUT/bench_k.json 2> $OUT/bench_k.err; echo "bench exit $?"
timeout 600 python bench.py --workload ml --no-cpu-baseline --no-row-sharded > $OUT/bench_ml_k.json 2> $OUT/bench_ml_k.err; echo "bench ml exit $?"
rm -rf /tmp/prof_k
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o bench -- python $OLDPWD/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded > $OLDPWD/$OUT/prof_k.log 2>&1; echo "prof exit $?")
DB=$(find /tmp/prof_k -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $OUT/bench_nf_kernel_stats_k.csv 121 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded (121 steps + 7 evaluations)"
python tools/step_timeline.py $DB $OUT/step_timeline_k.txt > /dev/null
bash tools/pmc_bench.sh $OUT/pmc_bench_step_k.json
python - <<'PY'
import json
for f in ('bench_k','bench_ml_k'):
    try:
        d=json.loads(open('gpurun_out/r03/%s.json'%f).read().strip().splitlines()[-1])
        print(f,{k:d[k] for k in ('value','ms_per_step')}, d['eval']['value'], d['parity']['ok'], d['roofline']['frac'], d['roofline']['traffic'], d.get('cpu_baseline'))
    except Exception as e: print(f,'ERR',e)
PY
