#!/bin/bash
# The evidence set of a round for the default bench workload: rocprofv3 kernel summary (--kernel-trace --stats) + one-step timeline, then the
# PMC passes over its step (tools/pmc_bench.sh). Usage: bash tools/profile_step.sh [out_dir]   (copy the results into profiles/rNN_*)
OUT=${1:-gpurun_out/prof}; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
rm -rf /tmp/prof_p
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o bench -- python $REPO/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded --no-end-to-end > $REPO/$OUT/prof_p.log 2>&1; echo "prof exit $?")
DB=$(find /tmp/prof_p -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $OUT/bench_nf_kernel_stats.csv 121 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded --no-end-to-end (121 steps + 7 evaluations)"
python tools/step_timeline.py $DB $OUT/step_timeline.txt > /dev/null
cut -c1-110 $OUT/step_timeline.txt
tail -1 $OUT/prof_p.log | cut -c1-200
bash tools/pmc_bench.sh $OUT/pmc_bench_step.json 2>&1 | tail -4 | cut -c1-300
