#!/bin/bash
# what separates the driver's 20-step / 5-warm-up reading from the default run's? (scratch tool)
for V in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --no-kernel-roofline" "--steps 20 --warmup 5 --no-kernel-roofline --no-parity" "--steps 20 --warmup 100" "--steps 200 --warmup 20"; do
  echo -n "$V: "
  python bench.py --gpus 1 $V --no-cpu-baseline --no-row-sharded --no-end-to-end --no-extra-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_hip_events'))"
done
