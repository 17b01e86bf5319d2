#!/bin/bash
# round-5 session 9: the non-default settings of the run-time switches against the default path
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_switches.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 15 $O/pytest.log
