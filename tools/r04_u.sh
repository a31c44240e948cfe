#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04u; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_trajectory.py -x -q -s -m gpu > $OUT/traj.log 2>&1; echo "traj rc $?"; grep "^\[trajectory" $OUT/traj.log | cut -c1-330; tail -1 $OUT/traj.log
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -1 $OUT/pytest.log | cut -c1-200
