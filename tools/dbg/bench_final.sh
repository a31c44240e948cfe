#!/bin/bash
# the round's final records: the default bench line (+ detail) and the driver's command (scratch tool)
python bench.py > gpurun_out/r06_bench_nf_final.json 2> gpurun_out/r06_bench_nf_final.err
cp bench_detail.json gpurun_out/r06_bench_nf_final_detail.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.json 2>/dev/null
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench_nf_final.json", "gpurun_out/r06_bench_driver_cmd.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d.get("ms_per_step_hip_events"), d["value"], d.get("eval"), (d.get("end_to_end") or {}).get("default"),
          ((d.get("spmm_roofline") or {}).get("ui") or {}).get("ms"), d.get("ml"), d.get("cfg5"), d.get("parity"))
PY
