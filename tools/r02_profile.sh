#!/bin/bash
# Round-2 profile pass: step time with the weight-gradients serial vs concurrent, rocprofv3 kernel trace of the bench
# (per-kernel stats CSV + one-step timeline). Output under gpurun_out/r02/.
TAG=${1:-v1}
OUT=gpurun_out/r02; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
for S in 1 0; do
  LLMREC_WGRAD_SERIAL=$S python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('wgrad_serial=$S', d['ms_per_step'], d['ms_per_step_hip_events'], d['eval']['ms'])"
done
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $REPO/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded > $REPO/$OUT/prof_$TAG.log 2>&1; echo "prof exit $?")
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
echo "db: $DB"
# 100 timed + 20 warm-up + 1 capture warm-up steps
python tools/rocpd_stats.py $DB $OUT/bench_nf_kernel_stats_$TAG.csv 121 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded (121 steps + 7 evaluations)"
python tools/step_timeline.py $DB $OUT/step_timeline_$TAG.txt
head -40 $OUT/bench_nf_kernel_stats_$TAG.csv
