#!/bin/bash
# Short GPU session for default-off experiments: A/B of the short bench (in-tree library against tools/abl/lib_<name>.so, per-kernel
# table: tools/gpu_ab.sh) and, if time is left, a few fast -m gpu tests THROUGH the experimental library.
#   usage: bash tools/gpu_exp.sh <tag> <name> [pytest -k expression]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r4x}; NAME=${2:-exp}; K=${3:-"mlp_fwd_bwd or render_vs_reference_golden"}
timeout 110 bash tools/gpu_ab.sh $TAG none default $NAME
L4D_LIB=$PWD/tools/abl/lib_$NAME.so timeout 45 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x --tb=line -k "$K" > gpurun_out/$TAG/pytest_$NAME.log 2>&1
echo "pytest ($NAME) rc=$?"; tail -n 4 gpurun_out/$TAG/pytest_$NAME.log
