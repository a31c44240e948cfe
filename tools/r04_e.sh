#!/bin/bash
# round 4, call E: suite (sync between graph executables), trajectories with the row-listed weight gradient, bench nf / ml
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py > gpurun_out/r04e/gpu_tests.log 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r04e/gpu_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_trajectory.py -x -q -s -m gpu > gpurun_out/r04e/traj.log 2>&1; echo "traj rc $?"; grep "^\[trajectory" gpurun_out/r04e/traj.log | cut -c1-330; tail -2 gpurun_out/r04e/traj.log
timeout 900 python bench.py --steps 200 --warmup 20 > gpurun_out/r04e/bench_nf.json 2> gpurun_out/r04e/bench_nf.err; echo "bench nf rc $?"; tail -3 gpurun_out/r04e/bench_nf.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04e/bench_nf.json").read().strip().splitlines()[-1])
    print("nf ms/step", d["ms_per_step"], "value", d["value"], "parity ok", d["parity"]["ok"])
    print({k:v for k,v in d["parity"].items() if k.endswith("rel") or k.endswith("_max") or "topk" in k and "note" not in k or "ulps" in k})
    r=d["roofline"]; print("roofline", r["ms_per_launch"], r["frac"], r.get("achieved"), "traffic", r.get("traffic"), "second", r["second"]["ms_per_launch"], r["second"]["frac"])
    print("wgrad rows", [k.get("rows") for k in d["kernels"] if "rows" in k])
    print("eval", d["eval"]); print("e2e", json.dumps(d.get("end_to_end"))[:1500])
    print("exact_f32", d.get("exact_f32",{}).get("ms_per_step"), "ref_order", d.get("reference_order",{}).get("ms_per_step"))
    print("spmm", json.dumps(d.get("spmm_roofline"))[:900])
    print("rs", {k:(v.get("ms_per_step"), v.get("row_restricted_forward",{}).get("ms_per_step")) for k,v in d.get("row_sharded",{}).items()})
    print("cpu", json.dumps(d.get("cpu_baseline"))[:600])
except Exception as e: print("no line", repr(e))
PY
timeout 600 python bench.py --workload ml --steps 200 --warmup 20 --no-cpu-baseline --no-row-sharded > gpurun_out/r04e/bench_ml.json 2> gpurun_out/r04e/bench_ml.err; echo "bench ml rc $?"
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04e/bench_ml.json").read().strip().splitlines()[-1])
    print("ml ms/step", d["ms_per_step"], "ok", d["parity"]["ok"], "ref_order", d.get("reference_order",{}).get("ms_per_step"), "rows", [k.get("rows") for k in d["kernels"] if "rows" in k])
except Exception as e: print("no line", repr(e))
PY
