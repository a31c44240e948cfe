#!/bin/bash
# the Python-driven row-sharded step (cfg-4 weak share) under rocprofv3: GPU busy fraction + per-kernel summary
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_q
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o rs -- python $OLDPWD/bench.py --workload cfg4 --synth-scaling weak --steps 20 --warmup 3 --no-parity --no-kernel-roofline > $OLDPWD/$OUT/prof_q.log 2>&1; echo "prof exit $?")
DB=$(find /tmp/prof_q -name "*.db" | head -1)
python tools/gpu_busy.py $DB sample | tee $OUT/row_sharded_gpu_busy.txt
python tools/rocpd_stats.py $DB $OUT/bench_cfg4_weak_kernel_stats.csv 24 "rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg4 --synth-scaling weak --steps 20 --warmup 3 --no-parity --no-kernel-roofline (24 steps: 3 warm-up + 20 timed + 1)"
head -12 $OUT/bench_cfg4_weak_kernel_stats.csv | cut -c1-150
