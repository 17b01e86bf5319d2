#!/bin/bash
# Kernel traces (rocprofv3 --kernel-trace --stats) of the other workloads' bench commands: gpurun_out/<tag>/kernel_stats_{c2,c5,c3_1k}.txt
#   usage: bash tools/gpu_traces_other.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r06}
O=gpurun_out/$TAG
mkdir -p $O
for W in c2 c5 c3-1k; do
  N=$(echo $W | tr '-' '_')
  B="python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt_$N -o kt -- $B > $GRAFT_REPO_ROOT/$O/kt_$N.log 2>&1 )
  python tools/rocpd_stats.py $(find $O/kt_$N -name "*.db" | head -1) 40 > $O/kernel_stats_$N.txt 2>&1
  head -4 $O/kernel_stats_$N.txt | cut -c1-150
  rm -rf $O/kt_$N $O/kt_$N.log
done
