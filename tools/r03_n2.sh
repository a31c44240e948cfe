#!/bin/bash
# plumbing check of `bench.py --gpus 2` (the N > 1 default: cfg 4 strong + single-GPU reference + Netflix replicas) on a 1-GPU box:
# two processes on cuda:0 over gloo. Not a performance number.
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp LLMREC_BENCH_SINGLE_DEVICE=1 LLMREC_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err; echo "n2 exit $?"
tail -c 1500 $OUT/bench_n2_gloo.json; tail -5 $OUT/bench_n2_gloo.err
