#!/bin/bash
# extra runs for the record: kernel probes, cfg 3 (MovieLens-shaped) bench, exact-fp32 projection bench
python tools/kernel_probe.py all 10
python bench.py --workload ml --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-roofline
LLMREC_GEMM=f32 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-roofline
