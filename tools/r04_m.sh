#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04m
for env in "LLMREC_REACH_STREAM=s2" "LLMREC_REACH_STREAM=s3" "LLMREC_WGRAD_BLOCKS=240" "LLMREC_WGRAD_BLOCKS=224"; do
  env $env LLMREC_SEGV_BT=$PWD/gpurun_out/r04m/segv.txt timeout 600 python bench.py --steps 200 --warmup 20 --no-end-to-end --no-cpu-baseline --no-row-sharded --no-kernel-roofline > gpurun_out/r04m/bench.json 2> gpurun_out/r04m/bench.err
  python - "$env" <<PY
import json,sys
try:
    d=json.loads(open("gpurun_out/r04m/bench.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "eval ms", round(d["eval"]["ms"],3), "parity", d["parity"]["ok"])
except Exception as e: print(sys.argv[1], "no line", repr(e)); print(open("gpurun_out/r04m/bench.err").read()[-600:])
PY
done
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -x -q -m gpu > gpurun_out/r04m/tests.log 2>&1; echo "ops+step rc $?"; tail -2 gpurun_out/r04m/tests.log | cut -c1-200
