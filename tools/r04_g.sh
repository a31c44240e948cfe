#!/bin/bash
# round 4, call G: native backtrace of the suite-order segfault without a debugger in the way
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04g
LLMREC_SEGV_BT=1 timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py -p no:faulthandler > gpurun_out/r04g/bt.log 2>&1; echo "suite (segv_bt, no faulthandler) rc $?"
grep -n "segv_bt" -A40 gpurun_out/r04g/bt.log | head -70 | cut -c1-240; tail -2 gpurun_out/r04g/bt.log | cut -c1-200
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py -p no:faulthandler > gpurun_out/r04g/nofh.log 2>&1; echo "suite (no faulthandler) rc $?"; tail -2 gpurun_out/r04g/nofh.log | cut -c1-200
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py > gpurun_out/r04g/fh.log 2>&1; echo "suite (default) rc $?"; tail -2 gpurun_out/r04g/fh.log | cut -c1-200
