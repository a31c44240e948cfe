#!/bin/bash
# The step's launch chain under rocprofv3 (--kernel-trace --stats), one run per variant of llmrec_amd/fused.py's knobs (LLMREC_FOLD,
# LLMREC_BWD_MAIN_FIRST, LLMREC_LOSS_STREAM, LLMREC_SPLIT_PROJ, LLMREC_SPLIT_WGRAD, LLMREC_ID_FIRST ...): a kernel summary (CSV), a one-step
# timeline, and the bench's own ms/step for each (same box). Usage: bash tools/step_variants.sh "<name>:<ENV=V ENV=V>" ...
OUT=${STEP_VARIANTS_OUT:-gpurun_out/r05chain}; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
ARGS="--steps 100 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded --no-end-to-end"
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  # plain timing first (no profiler)
  env $envs python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded --no-end-to-end > $OUT/$name.bench.json 2> $OUT/$name.bench.err
  python -c "import json;d=json.load(open('$OUT/$name.bench.json'));print('$name', 'ms_per_step', round(d['ms_per_step'],4), 'eval_ms', round(d['eval']['ms'],4))"
  rm -rf /tmp/prof_$name
  (cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o bench -- python $REPO/bench.py $ARGS > $REPO/$OUT/$name.prof.log 2>&1; echo "$name prof exit $?")
  DB=$(find /tmp/prof_$name -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB $OUT/$name.kernel_stats.csv 121 "rocprofv3 --kernel-trace --stats -- $envs python bench.py $ARGS (121 steps + 7 evaluations)" > /dev/null
  python tools/step_timeline.py $DB $OUT/$name.timeline.txt > /dev/null
  tail -1 $OUT/$name.timeline.txt
done
