#!/bin/bash
# round 4, call A: trajectory parity, launcher tests, the whole GPU suite, python main.py end to end
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_bench_launch.py -x -q -s -m gpu > gpurun_out/r04a/traj.log 2>&1; echo "traj rc $?"
grep "^\[trajectory" gpurun_out/r04a/traj.log; tail -5 gpurun_out/r04a/traj.log
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py > gpurun_out/r04a/gpu_tests.log 2>&1; echo "suite rc $?"
tail -4 gpurun_out/r04a/gpu_tests.log
timeout 1200 python tools/e2e_main.py --epochs 6 --out gpurun_out/r04a/e2e_main.json > gpurun_out/r04a/e2e.log 2>&1; echo "e2e rc $?"
grep "^\[e2e\]" gpurun_out/r04a/e2e.log | cut -c1-900
