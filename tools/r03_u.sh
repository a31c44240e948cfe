#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k "listed_rows_and_of_needed" 2>&1 | grep -E "^E |passed|failed" | head -12 | cut -c1-250
