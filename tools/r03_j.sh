#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x -k "topk or eval or score or step or candidates or metrics" 2>&1 | tail -4 | tee $OUT/pytest_j.txt
timeout 120 python tools/kernel_probe.py topk 20 2>&1 | grep score_topk | tee $OUT/topk_j.txt
timeout 300 python bench.py --no-cpu-baseline --no-row-sharded --no-kernel-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],4), 'eval', d['eval'], d['parity']['ok'], d['parity']['topk_lists_equal'])" | tee -a $OUT/topk_j.txt
