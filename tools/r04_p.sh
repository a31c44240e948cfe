#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04p
for env in "LLMREC_PREPROPAGATE=0 LLMREC_WGRAD_BLOCKS=0" "LLMREC_PREPROPAGATE=0 LLMREC_WGRAD_BLOCKS=224" "LLMREC_PREPROPAGATE=0 LLMREC_WGRAD_BLOCKS=240"; do
  env $env timeout 600 python bench.py --steps 200 --warmup 20 --no-end-to-end --no-cpu-baseline --no-row-sharded --no-kernel-roofline --no-parity > gpurun_out/r04p/bench.json 2> gpurun_out/r04p/bench.err
  python - "$env" <<PY
import json,sys
try:
    d=json.loads(open("gpurun_out/r04p/bench.json").read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "eval ms", round(d["eval"]["ms"],3))
except Exception as e: print(sys.argv[1], "no line", repr(e))
PY
done
for env in "LLMREC_WGRAD_BLOCKS=0" "LLMREC_WGRAD_BLOCKS=224"; do
env $env timeout 600 python bench.py --workload ml --steps 200 --warmup 20 --no-cpu-baseline --no-row-sharded > gpurun_out/r04p/bench_ml.json 2> gpurun_out/r04p/bench_ml.err; echo "bench ml rc $?"
python - "$env" <<PY
import json,sys
try:
    d=json.loads(open("gpurun_out/r04p/bench_ml.json").read().strip().splitlines()[-1])
    print("ml", sys.argv[1], "ms/step", d["ms_per_step"], "ok", d["parity"]["ok"], "other", d.get("pre_propagated_order",{}).get("ms_per_step"), d.get("reference_order",{}).get("ms_per_step"), "eval", d["eval"]["ms"])
except Exception as e: print("no line", repr(e))
PY
done
