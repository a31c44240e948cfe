#!/bin/bash
# round 4, call D: suite with per-test release of graph executables, wgrad row-list microbench, SpMM PMC at 40 M, e2e after the device_state fix
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py > gpurun_out/r04d/gpu_tests.log 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r04d/gpu_tests.log | cut -c1-300
timeout 300 python tools/wgrad_rows_probe.py > gpurun_out/r04d/wgrad_rows_probe.log 2>&1; echo "probe rc $?"; cat gpurun_out/r04d/wgrad_rows_probe.log | tail -16
bash tools/pmc_spmm.sh gpurun_out/r04d/pmc_spmm_40M.json 2>&1 | tail -5
timeout 900 python tools/e2e_main.py --epochs 6 --modes default,graph_device_sampler --out gpurun_out/r04d/e2e_main.json > gpurun_out/r04d/e2e.log 2>&1; echo "e2e rc $?"
grep "^\[e2e\]" gpurun_out/r04d/e2e.log | cut -c1-700
