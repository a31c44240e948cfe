#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
for V in 2 4; do LLMREC_WGRAD_KERNEL=$V timeout 120 python tools/wgrad_probe.py 30 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-400; done | tee $OUT/wgrad_probe_d.txt
LLMREC_WGRAD_KERNEL=4 timeout 600 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_ops.py tests/test_gpu_step.py -q -x -k "wgrad or weight or linear or step" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_d.txt
LLMREC_WGRAD_KERNEL=4 timeout 300 python bench.py --no-cpu-baseline --no-row-sharded --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('V=4 ms/step', round(d['ms_per_step'],4), 'roofline', round(d['roofline']['frac'],3), [ (k['kernel'][:30], round(k['ms'],4)) for k in d['kernels'][:2]])" | tee $OUT/bench_d.txt
