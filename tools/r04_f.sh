#!/bin/bash
# round 4, call F: native backtrace of the suite-order segfault
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04f
export LLMREC_TRACE_CAPTURE=1
timeout 1200 rocgdb -q -batch -ex "set pagination off" -ex "set confirm off" -ex "handle SIGSEGV stop nopass" -ex run -ex "bt 40" -ex "info threads" -ex "thread apply all bt 12" \
  --args python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py -p no:faulthandler > gpurun_out/r04f/gdb.log 2>&1; echo "gdb rc $?"
grep -n "SIGSEGV" -A60 gpurun_out/r04f/gdb.log | head -120 | cut -c1-260
grep "^\[capture\]" gpurun_out/r04f/gdb.log | tail -8
