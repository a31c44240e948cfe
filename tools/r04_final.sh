#!/bin/bash
# what the driver runs at round end, in the same order: the whole GPU suite, smoke(), the default bench line; then the evidence set
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04final; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_final.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest_final.log | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | cut -c1-300
/usr/bin/time -f "bench wall %e s" timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench exit $?"; tail -1 $OUT/bench_driver_cmd.err
timeout 1200 python bench.py > $OUT/bench_nf_final.json 2> $OUT/bench_nf_final.err; echo "bench (defaults) exit $?"
python - <<'PY'
import json
for f in ("bench_driver_cmd","bench_nf_final"):
    try:
        d=json.loads(open('gpurun_out/r04final/%s.json'%f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f, {k:d[k] for k in ('value','ms_per_step','steps','warmup')}, 'eval', d['eval']['value'], 'parity', d['parity']['ok'], 'roof', r['kernel'][:30], round(r['frac'],3), r['traffic'], 'second', round(r['second']['frac'],3),
              'cpu', d['cpu_baseline']['value'], 'e2e', {m:(round(d['end_to_end'][m]['edges_per_s']), round(d['end_to_end'][m]['users_per_s'])) for m in ('default','graph_device_sampler')} if 'end_to_end' in d and 'default' in d['end_to_end'] else d.get('end_to_end'),
              'rs', {k:(round(v.get('ms_per_step',0),2), round(v.get('row_restricted_forward',{}).get('ms_per_step',0),2)) for k,v in d.get('row_sharded',{}).items()})
    except Exception as e: print(f, 'no line', repr(e))
PY
bash tools/r04_prof.sh 2>&1 | tail -45
