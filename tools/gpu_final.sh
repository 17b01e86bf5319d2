#!/bin/bash
# One GPU session for a finished build: the whole -m gpu suite; an A/B of the short bench against tools/abl/lib_base.so (the
# previous build) with the per-kernel table; and -- only if the suite is green -- the profile round (tools/gpu_profile_round.sh
# <tag> skip-tests: kernel trace, PMC passes, calibration, bench lines of every workload).
#   usage: bash tools/gpu_final.sh <tag> [no-ab]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r04}
O=gpurun_out/${TAG}_final
mkdir -p $O
sha256sum lidar4d_amd/liblidar4d_hip.so | cut -c1-16 > $O/lib_sha.txt
timeout 720 python -m pytest tests -m gpu -q -rfE --tb=short > $O/pytest.log 2>&1
RC=$?
echo "pytest rc=$RC" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error|rc=" $O/pytest.log | tail -n 12
[ "$2" == "no-ab" ] || bash tools/gpu_ab.sh ${TAG}_ab none base default
if [ $RC -ne 0 ]; then
  echo "suite not green: no profile round"
  tail -n 60 $O/pytest.log
  exit 1
fi
bash tools/gpu_profile_round.sh $TAG skip-tests
