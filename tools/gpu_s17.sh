#!/bin/bash
# round-5 session 17: pass 1 of the sorted scatter with two barriers per level (copy-out under the bin scan), and the library linked
# with a version script (only l4d_* exported): tests + A/B against the shipped build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s17; mkdir -p $O
L4D_LIB=$PWD/tools/abl/lib_p1b2.so timeout 300 python -m pytest tests/test_gpu_properties.py tests/test_gpu_ops.py -m gpu -q --tb=short -k "scatter or hashgrid or binned" > $O/pytest_p1b2.log 2>&1; echo "pytest p1b2 rc=$?"; tail -n 2 $O/pytest_p1b2.log
L4D_LIB=$PWD/tools/abl/lib_vis.so timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -x > $O/pytest_vis.log 2>&1; echo "pytest vis rc=$?"; tail -n 2 $O/pytest_vis.log
bash tools/gpu_ab.sh s17 none base p1b2 vis
