#!/bin/bash
# the gradient-yardstick test alone, with its printed table (hash tables + the 24 hex-planes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4y
python -m pytest tests/test_gpu_c3_parity.py -m gpu -q -s -k "yardstick" > gpurun_out/r4y/yardstick.txt 2>&1
grep -E "HIP |passed|failed" gpurun_out/r4y/yardstick.txt | cut -c1-230
