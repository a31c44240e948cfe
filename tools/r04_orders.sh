#!/bin/bash
# robustness of the GPU_MAX_HW_QUEUES work-around: the same tests in other orders (a different history of stream creations before every capture)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04u; mkdir -p $OUT
FILES=$(ls tests/test_gpu_*.py | sort -r | tr '\n' ' ')
timeout 1500 python -m pytest $FILES -x -q -m gpu -p no:cacheprovider > $OUT/reverse.log 2>&1; echo "reverse file order rc $?"; tail -1 $OUT/reverse.log | cut -c1-200
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_bench_shapes.py tests/test_gpu_step.py tests/test_gpu_options.py tests/test_gpu_dp_multiproc.py tests/test_gpu_step.py -x -q -m gpu -p no:cacheprovider > $OUT/repeat.log 2>&1; echo "step x3 interleaved rc $?"; tail -1 $OUT/repeat.log | cut -c1-200
