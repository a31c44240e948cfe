#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04h
LLMREC_SEGV_BT=1 timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py -p no:faulthandler > gpurun_out/r04h/bt.log 2>&1; echo "suite (segv_bt in run_steps) rc $?"
grep -n "segv_bt" -A45 gpurun_out/r04h/bt.log | head -80 | cut -c1-250; tail -c 600 gpurun_out/r04h/bt.log
setarch -R true 2>/dev/null && { setarch -R timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py > gpurun_out/r04h/noaslr.log 2>&1; echo "suite (ASLR off) rc $?"; tail -2 gpurun_out/r04h/noaslr.log | cut -c1-200; }
LLMREC_TRACE_CAPTURE=1 timeout 900 python -m pytest tests -x -q -m gpu -s --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py > gpurun_out/r04h/trace.log 2>&1; echo "suite (trace, -s) rc $?"; grep "^\[capture\]" gpurun_out/r04h/trace.log | tail -6 | cut -c1-250; tail -c 300 gpurun_out/r04h/trace.log
