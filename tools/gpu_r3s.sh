#!/bin/bash
# round 3, run S: the full -m gpu suite on the final build (incl. the replay-equals-eager regression test)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3s}
mkdir -p $O
sha256sum lidar4d_amd/liblidar4d_hip.so | cut -c1-16 > $O/pytest_gpu_final.txt
timeout 420 python -m pytest tests -m gpu -q -rfE --tb=short >> $O/pytest_gpu_final.txt 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_final.txt
tail -n 12 $O/pytest_gpu_final.txt | cut -c1-300
