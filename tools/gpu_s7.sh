#!/bin/bash
# round-5 session 7: persistent wavefronts in the fused encode + density network kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_properties.py -m gpu -q -x --tb=short -k "epilogue or level_major_flow" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -n 3 $O/pytest_new.log
timeout 600 bash tools/gpu_ab.sh s7 none persist onegroup:L4D_ENC_PERSISTENT=0 nosigma:L4D_ENC_SIGMA=0 norecomp:L4D_MLP_RECOMP_SIGMA=0
