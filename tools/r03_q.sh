#!/bin/bash
OUT=gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_q
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_q -o rs -- python $OLDPWD/bench.py --workload cfg4 --synth-scaling weak --steps 20 --warmup 3 --no-parity > $OLDPWD/$OUT/prof_q.log 2>&1; echo "prof exit $?")
DB=$(find /tmp/prof_q -name "*.db" | head -1)
python tools/gpu_busy.py $DB sample | tee $OUT/row_sharded_gpu_busy.txt
tail -c 400 $OUT/prof_q.log
