#!/bin/bash
# One GPU session: parity tests, smoke, bench (+ rocprofv3 kernel stats); logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 50 --warmup 10 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
rm -rf gpurun_out/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-roofline > $REPO/gpurun_out/prof.log 2>&1; echo "prof exit $?" >> $REPO/gpurun_out/prof.log)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --workload synth --steps 10 --warmup 3 > gpurun_out/bench_synth.log 2>&1; echo "synth exit $?" >> gpurun_out/bench_synth.log
[ -f tools/extra_round.sh ] && bash tools/extra_round.sh > gpurun_out/extra.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -c 400 gpurun_out/bench.log; tail -c 700 gpurun_out/bench_synth.log; tail -2 gpurun_out/prof.log
