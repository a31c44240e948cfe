#!/bin/bash
# round 4, call C: bisect the suite-order segfault in test_in_graph_sampler_epoch_sums_on_the_device
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04c
T=tests/test_gpu_step.py::test_in_graph_sampler_epoch_sums_on_the_device
for f in options ops dist_fused dp_multiproc bench_shapes; do
  timeout 600 python -m pytest tests/test_gpu_$f.py $T -x -q -m gpu > gpurun_out/r04c/pair_$f.log 2>&1; echo "pair $f rc $?"
done
LLMREC_GRAPH=0 timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_trajectory.py --deselect tests/test_gpu_bench_launch.py > gpurun_out/r04c/suite_graph0.log 2>&1; echo "suite LLMREC_GRAPH=0 rc $?"; tail -2 gpurun_out/r04c/suite_graph0.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_options.py tests/test_gpu_step.py -x -q -m gpu > gpurun_out/r04c/options_step.log 2>&1; echo "options+step rc $?"; tail -2 gpurun_out/r04c/options_step.log | cut -c1-200
timeout 600 python tools/eval_probe.py > gpurun_out/r04c/eval_probe.log 2>&1; echo "probe rc $?"; grep " ms\|cumtime\|tottime" -A0 gpurun_out/r04c/eval_probe.log | head -60
