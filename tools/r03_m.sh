#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-row-sharded --no-kernel-roofline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['eval']['ms'], d['eval']['value'])"
done
