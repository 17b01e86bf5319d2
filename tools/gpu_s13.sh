#!/bin/bash
# round-5 session 13: the flow network's backward at two wavefronts per SIMD (no next-tile prefetch, two workgroups per CU)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 bash tools/gpu_ab.sh s13 none w2:L4D_MLP_BWD_GRID_NARROW=2 w2grid1 base
timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --tb=short -k "mlp" > gpurun_out/s13/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/s13/pytest.log
