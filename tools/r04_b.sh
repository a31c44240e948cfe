#!/bin/bash
# round 4, call B: row-listed weight gradient (tests, A/B), suite re-run (was the round-A segfault a flake?), eval probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04b
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "reach or row_list or weight_gradient" > gpurun_out/r04b/new_ops.log 2>&1; echo "new ops rc $?"; tail -3 gpurun_out/r04b/new_ops.log
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py > gpurun_out/r04b/gpu_tests.log 2>&1; echo "suite rc $?"; tail -4 gpurun_out/r04b/gpu_tests.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_step.py -x -q -m gpu > gpurun_out/r04b/step_tests.log 2>&1; echo "step tests rc $?"; tail -2 gpurun_out/r04b/step_tests.log | cut -c1-300
timeout 600 python tools/eval_probe.py > gpurun_out/r04b/eval_probe.log 2>&1; echo "probe rc $?"; grep " ms " gpurun_out/r04b/eval_probe.log
for rows in 1 0; do
  LLMREC_WGRAD_ROWS=$rows timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-row-sharded > gpurun_out/r04b/bench_nf_rows$rows.json 2> gpurun_out/r04b/bench_nf_rows$rows.err; echo "bench rows=$rows rc $?"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04b/bench_nf_rows$rows.json").read().strip().splitlines()[-1])
    print("rows=$rows ms/step", d["ms_per_step"], "ok", d["parity"]["ok"], "roof", d["roofline"]["ms_per_launch"], d["roofline"]["frac"], "ref_order", d.get("reference_order",{}).get("ms_per_step"), "eval", d["eval"]["ms"])
    print({k:v for k,v in d["parity"].items() if k.endswith("rel") or "topk" in k})
except Exception as e: print("no line", e)
PY
done
LLMREC_WGRAD_ROWS=1 timeout 600 python bench.py --workload ml --steps 200 --warmup 20 --no-cpu-baseline --no-row-sharded > gpurun_out/r04b/bench_ml.json 2> gpurun_out/r04b/bench_ml.err; echo "bench ml rc $?"
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04b/bench_ml.json").read().strip().splitlines()[-1])
    print("ml ms/step", d["ms_per_step"], "ok", d["parity"]["ok"], "ref_order", d.get("reference_order",{}).get("ms_per_step"))
except Exception as e: print("no line", e)
PY
