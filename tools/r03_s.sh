#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dist_fused.py tests/test_gpu_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "spmm or sharded or dist or row or mark or listed" 2>&1 | tail -4 | cut -c1-300
for S in weak strong; do for F in "" "--synth-dense-forward" "--synth-dense-backward"; do
  echo "== cfg4 $S $F"; timeout 600 python bench.py --workload cfg4 --synth-scaling $S --steps 10 --warmup 2 --no-kernel-roofline --no-parity $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['loss'], d['mf_emb'], d['messages'].get('exchanged_bytes_last_step'))"
done; done 2>&1 | tee $OUT/sparse_bwd_ab.txt
