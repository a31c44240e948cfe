#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
for V in 2 8; do for A in 0 3 4; do LLMREC_LIB=$PWD/llmrec_amd/lib/libllmrec_hip_tools.so LLMREC_WGRAD_KERNEL=$V LLMREC_WGRAD_ABL=$A timeout 120 python tools/wgrad_probe.py 10 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-160; done; done | tee $OUT/wgrad_clock_e4.txt
