#!/bin/bash
# round 4, call I: suite with GC held off during captures (twice), the evidence set (rocprofv3 summary, timeline, PMC), ML line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04i
for k in 1 2; do
  timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py > gpurun_out/r04i/gpu_tests_$k.log 2>&1; echo "suite $k rc $?"; tail -2 gpurun_out/r04i/gpu_tests_$k.log | cut -c1-250
done
bash tools/r04_prof.sh 2>&1 | tail -60
cp gpurun_out/r04prof/* gpurun_out/r04i/ 2>/dev/null
timeout 600 python bench.py --workload ml --steps 200 --warmup 20 --no-cpu-baseline --no-row-sharded > gpurun_out/r04i/bench_ml.json 2> gpurun_out/r04i/bench_ml.err; echo "bench ml rc $?"
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04i/bench_ml.json").read().strip().splitlines()[-1])
    print("ml ms/step", d["ms_per_step"], "ok", d["parity"]["ok"], "other", d.get("pre_propagated_order",{}).get("ms_per_step"), d.get("reference_order",{}).get("ms_per_step"), "eval", d["eval"]["ms"])
    print({k:v for k,v in d["parity"].items() if k.endswith("rel") or k.endswith("_max") or "ulps" in k or k.startswith("topk_lists")})
except Exception as e: print("no line", repr(e))
PY
