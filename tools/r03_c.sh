#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
for A in 0 1 2 3 4 5 6 7; do LLMREC_WGRAD_KERNEL=2 LLMREC_WGRAD_ABL=$A timeout 300 python tools/wgrad_probe.py 30 2>&1 | grep WGRAD | cut -c1-130; done | tee $OUT/wgrad_abl_c.txt
