#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04n
for env in "GPU_MAX_HW_QUEUES=8" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8"; do
  env $env timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py > gpurun_out/r04n/suite.log 2>&1; echo "suite $env rc $?"; tail -1 gpurun_out/r04n/suite.log | cut -c1-200
done
for env in "LLMREC_WGRAD_BLOCKS=208" "LLMREC_WGRAD_BLOCKS=192" "GPU_MAX_HW_QUEUES=8" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8"; do
  env $env timeout 600 python bench.py --steps 200 --warmup 20 --no-end-to-end --no-cpu-baseline --no-row-sharded --no-kernel-roofline --no-parity > gpurun_out/r04n/bench.json 2> gpurun_out/r04n/bench.err
  python - "$env" <<PY
import json,sys
try:
    d=json.loads(open("gpurun_out/r04n/bench.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "eval ms", round(d["eval"]["ms"],3))
except Exception as e: print(sys.argv[1], "no line", repr(e))
PY
done
