#!/bin/bash
# rocprofv3 kernel summary + one-step timeline of the default bench workload
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_p
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o bench -- python $OLDPWD/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded > $OLDPWD/$OUT/prof_p.log 2>&1; echo "prof exit $?")
DB=$(find /tmp/prof_p -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $OUT/bench_nf_kernel_stats_p.csv 121 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded (121 steps + 7 evaluations)"
python tools/step_timeline.py $DB $OUT/step_timeline_p.txt > /dev/null
cut -c1-100 $OUT/step_timeline_p.txt
tail -1 $OUT/prof_p.log | cut -c1-200
