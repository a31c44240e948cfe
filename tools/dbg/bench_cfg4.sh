#!/bin/bash
# BASELINE.json configs[3] whole on one GPU (scratch tool: the record of profiles/r06_bench_cfg4_full_1gpu.json)
python bench.py --workload cfg4 --synth-scaling strong --steps 10 --warmup 2 > gpurun_out/r06_bench_cfg4_full_1gpu.json 2> gpurun_out/cfg4.err
cp bench_detail.json gpurun_out/r06_bench_cfg4_full_1gpu_detail.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_cfg4_full_1gpu.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], {k: v for k, v in d.items() if "eval" in k})
PY
