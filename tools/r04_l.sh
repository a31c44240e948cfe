#!/bin/bash
# round 4, call L: HIP's hardware-queue pool vs the parallel-stream assignment of hipGraphLaunch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04l
for q in 8 16; do
  GPU_MAX_HW_QUEUES=$q LLMREC_SEGV_BT=$PWD/gpurun_out/r04l/segv_q$q.txt timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py > gpurun_out/r04l/suite_q$q.log 2>&1; echo "suite GPU_MAX_HW_QUEUES=$q rc $?"; tail -1 gpurun_out/r04l/suite_q$q.log | cut -c1-200
done
DEBUG_HIP_FORCE_GRAPH_QUEUES=2 timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py > gpurun_out/r04l/suite_fgq2.log 2>&1; echo "suite DEBUG_HIP_FORCE_GRAPH_QUEUES=2 rc $?"; tail -1 gpurun_out/r04l/suite_fgq2.log | cut -c1-200
for env in "X=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "LLMREC_WGRAD_BLOCKS=240" "LLMREC_WGRAD_BLOCKS=224"; do
  env $env timeout 600 python bench.py --steps 200 --warmup 20 --no-end-to-end --no-cpu-baseline --no-row-sharded --no-kernel-roofline --no-parity > gpurun_out/r04l/bench.json 2> gpurun_out/r04l/bench.err
  python - "$env" <<PY
import json,sys
try:
    d=json.loads(open("gpurun_out/r04l/bench.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "eval ms", round(d["eval"]["ms"],3))
except Exception as e: print(sys.argv[1], "no line", repr(e))
PY
done
