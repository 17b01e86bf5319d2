#!/bin/bash
# round-5 session 1: gather cache-policy microbenchmark + A/B of the three default-off knobs left by round 4
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
timeout 120 tools/ubench/gather_policy_ubench > $O/ubench_gather_policy.txt 2>&1; echo "ubench rc=$?"
timeout 400 bash tools/gpu_ab.sh s1 none default pref wscan hdt
for n in wscan pref; do
  L4D_LIB=$PWD/tools/abl/lib_$n.so timeout 200 python -m pytest tests/test_gpu_properties.py tests/test_gpu_ops.py -m gpu -q -x --tb=line -k "scatter or binned or plane or reproduc" > $O/pytest_$n.log 2>&1
  echo "pytest $n rc=$?"; tail -n 3 $O/pytest_$n.log
done
