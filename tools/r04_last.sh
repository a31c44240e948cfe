#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04last; mkdir -p $OUT
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -1 $OUT/pytest.log | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 900 python tools/e2e_main.py --epochs 8 --modes default,graph_device_sampler --out $OUT/e2e_main.json > $OUT/e2e.log 2>&1; echo "e2e rc $?"
grep "^\[e2e\]" $OUT/e2e.log | cut -c1-520
timeout 600 python bench.py --steps 200 --warmup 20 --no-end-to-end --no-cpu-baseline --no-row-sharded --no-kernel-roofline > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['parity']['ok'])"
