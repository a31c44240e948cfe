#!/bin/bash
# DP (batch-sharded replicas) checks on a 1-GPU box: new parity tests, the 3-segment step with and without RCCL calls.
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "data_parallel or multi_sharded" > gpurun_out/pytest_dp.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_dp.log
LLMREC_FORCE_DP=1 timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-roofline > gpurun_out/bench_dp1.log 2>&1; echo "exit $?" >> gpurun_out/bench_dp1.log
LLMREC_FORCE_DP=1 LLMREC_DP_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-roofline > gpurun_out/bench_dp1_rccl.log 2>&1; echo "exit $?" >> gpurun_out/bench_dp1_rccl.log
tail -5 gpurun_out/pytest_dp.log; tail -c 900 gpurun_out/bench_dp1.log; tail -c 900 gpurun_out/bench_dp1_rccl.log
