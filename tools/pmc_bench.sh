#!/bin/bash
# PMC passes over the bench's own training step (eager launches of the same kernels the graph replays), separate
# rocprofv3 runs with --kernel-trace only (MI355X_MICROARCH.md). Output: per-kernel means -> profiles/r02_pmc_bench_step.json
OUT=${1:-gpurun_out/pmc_bench_step.json}
REPO=$PWD; export TMPDIR=/tmp
rm -rf /tmp/pmc_runs; mkdir -p /tmp/pmc_runs gpurun_out
[ -n "$PMC_KEEP" ] && mkdir -p /tmp/pmc_runs/keep && cp $PMC_KEEP/*counter_collection.csv /tmp/pmc_runs/keep/ 2>/dev/null
i=0
# FETCH_SIZE costs 3 of the 4 TCC slots and WRITE_SIZE 2: separate passes (together the run never finishes)
SETS=${PMC_SETS:-"FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum|SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"}
IFS="|" read -ra ARR <<< "$SETS"
for SET in "${ARR[@]}"; do
  i=$((i+1))
  (cd /tmp && LLMREC_GRAPH=0 timeout 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_runs/p$i -o run -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-roofline --no-parity --no-row-sharded --no-end-to-end > /tmp/pmc_runs/p$i.log 2>&1; echo "pass $i ($SET) exit $?")
done
python $REPO/tools/pmc_aggregate.py /tmp/pmc_runs $OUT | cut -c1-400
