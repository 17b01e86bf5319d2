#!/bin/bash
# round-5 session 15: pass 1 of the sorted scatter with 1,024-thread tiles (tools/abl/lib_bs1024.so): scatter tests + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s15; mkdir -p $O
L4D_LIB=$PWD/tools/abl/lib_bs1024.so timeout 300 python -m pytest tests/test_gpu_properties.py tests/test_gpu_ops.py -m gpu -q --tb=short -k "scatter or hashgrid or binned" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $O/pytest.log
bash tools/gpu_ab.sh s15 none default bs1024
