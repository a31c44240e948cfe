#!/bin/bash
# One GPU session: parity tests, smoke, bench (+ rocprofv3 kernel stats); logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
REPO=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $REPO/gpurun_out/prof.log 2>&1; echo "prof exit $?" >> $REPO/gpurun_out/prof.log)
find gpurun_out/prof -name "*stats*" | head; find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
timeout 600 python bench.py --workload synth --steps 10 --warmup 3 > gpurun_out/bench_synth.log 2>&1; echo "synth exit $?" >> gpurun_out/bench_synth.log
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -c 600 gpurun_out/bench.log; tail -c 900 gpurun_out/bench_synth.log; tail -3 gpurun_out/prof.log
