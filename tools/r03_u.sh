#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dist_fused.py -m gpu -q -x -p no:cacheprovider -k "two_processes" 2>&1 | grep -E "Error|error|assert|rank|Traceback|line " | head -40 | cut -c1-250
