#!/bin/bash
# round 4, call K: graphs replayed on a high-priority launch stream (HIP's parallel-stream assignment bug): suite twice, trajectories, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04k
for k in 1 2; do
  LLMREC_SEGV_BT=$PWD/gpurun_out/r04k/segv_bt_$k.txt timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py > gpurun_out/r04k/gpu_tests_$k.log 2>&1; echo "suite $k rc $?"; tail -2 gpurun_out/r04k/gpu_tests_$k.log | cut -c1-250
done
head -8 gpurun_out/r04k/segv_bt_1.txt 2>/dev/null | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_trajectory.py -x -q -s -m gpu > gpurun_out/r04k/traj.log 2>&1; echo "traj rc $?"; grep "^\[trajectory" gpurun_out/r04k/traj.log | cut -c1-330; tail -1 gpurun_out/r04k/traj.log
timeout 900 python bench.py --steps 200 --warmup 20 --no-end-to-end --no-cpu-baseline --no-row-sharded > gpurun_out/r04k/bench_nf.json 2> gpurun_out/r04k/bench_nf.err; echo "bench nf rc $?"
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r04k/bench_nf.json").read().strip().splitlines()[-1])
    print("nf ms/step", d["ms_per_step"], "events", d["ms_per_step_hip_events"], "parity ok", d["parity"]["ok"], "eval ms", d["eval"]["ms"])
    print({k:v for k,v in d["parity"].items() if "ulps" in k or k.startswith("topk_m")})
except Exception as e: print("no line", repr(e))
PY
