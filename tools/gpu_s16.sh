#!/bin/bash
# round-5 session 16: 256-thread tiles in pass 1 of the sorted scatter; 8-lane merge width in the plane adjoints
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s16; mkdir -p $O
for v in bs256 pdyn8 pstat8; do
  L4D_LIB=$PWD/tools/abl/lib_$v.so timeout 300 python -m pytest tests/test_gpu_properties.py tests/test_gpu_ops.py -m gpu -q --tb=short -k "scatter or hashgrid or binned or band or planes" > $O/pytest_$v.log 2>&1; echo "pytest $v rc=$?"; tail -n 2 $O/pytest_$v.log
done
bash tools/gpu_ab.sh s16 none default bs256 pdyn8 pstat8
