#!/bin/bash
# round-5 session 8: non-temporal streams in the fused encode and the time-plane kernel; segment bounds without scratch
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
timeout 600 bash tools/gpu_ab.sh s8 none nt prent
timeout 900 python -m pytest tests/test_gpu_c3_parity.py tests/test_gpu_properties.py tests/test_gpu_model.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
