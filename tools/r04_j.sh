#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04j
rm -f gpurun_out/r04j/segv_bt.txt
LLMREC_SEGV_BT=$PWD/gpurun_out/r04j/segv_bt.txt timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py -p no:faulthandler > gpurun_out/r04j/suite.log 2>&1; echo "suite rc $?"
tail -c 300 gpurun_out/r04j/suite.log; echo; cat gpurun_out/r04j/segv_bt.txt 2>/dev/null | cut -c1-220 | head -80
