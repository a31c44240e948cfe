#!/bin/bash
python tools/kernel_probe.py fwd 20
python tools/kernel_probe.py wgrad 20
LLMREC_GEMM=f32 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-roofline
