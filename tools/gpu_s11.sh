#!/bin/bash
# round-5 session 11: the tests added after the last full run, on the final library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
sha256sum lidar4d_amd/liblidar4d_hip.so | cut -c1-16
timeout 400 python -m pytest tests/test_gpu_properties.py -m gpu -q --tb=short -k "level_major or epilogue" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
