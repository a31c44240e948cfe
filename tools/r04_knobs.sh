#!/bin/bash
# runtime knobs vs the step time (bench, 200 steps, no extras)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04knobs
for env in "X=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=3" "GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=6" "X=2"; do
  env $env timeout 600 python bench.py --steps 400 --warmup 40 --no-end-to-end --no-cpu-baseline --no-row-sharded --no-kernel-roofline --no-parity > gpurun_out/r04knobs/bench.json 2> gpurun_out/r04knobs/bench.err
  python - "$env" <<PY
import json,sys
try:
    d=json.loads(open("gpurun_out/r04knobs/bench.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "eval ms", round(d["eval"]["ms"],3))
except Exception as e: print(sys.argv[1], "no line", repr(e))
PY
done
