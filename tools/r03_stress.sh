#!/bin/bash
# repeat the GPU suite to catch intermittent failures (races in the fused launches, graph capture order)
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1
done | tee $OUT/stress.txt
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-row-sharded --no-kernel-roofline --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parity']['ok'], d['parity']['embeddings_after_steps_max_rel'], d['parity']['grad_max_rel'])"; done | tee -a $OUT/stress.txt
