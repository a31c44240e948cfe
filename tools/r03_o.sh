#!/bin/bash
# the full GPU suite, then cfg 4 and cfg 5 whole on one GPU (exact edge counts), then the N = 2 plumbing check
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
timeout 900 python bench.py --workload cfg4 --synth-scaling strong --steps 10 --warmup 2 > $OUT/bench_cfg4_o.json 2> $OUT/bench_cfg4_o.err; echo "cfg4 exit $?"
timeout 1500 python bench.py --workload cfg5 --synth-scaling strong > $OUT/bench_cfg5_o.json 2> $OUT/bench_cfg5_o.err; echo "cfg5 exit $?"; tail -2 $OUT/bench_cfg5_o.err | cut -c1-200
bash tools/r03_n2.sh 2>&1 | tail -1 | cut -c1-200
python - <<'PY'
import json
for f in ('bench_cfg4_o','bench_cfg5_o','bench_n2_gloo'):
    try:
        d=json.loads(open('gpurun_out/r03/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['value'], d.get('propagated_edges_per_sec'), d.get('parity'), d.get('ingest',{}).get('hbm_peak_gb'))
    except Exception as e: print(f,'ERR',e)
PY
