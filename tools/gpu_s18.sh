#!/bin/bash
# round-5 session 18: rocprofv3 kernel traces of the two other single-GPU workloads (C5 inference + U-Net + chamfer evaluation, C2) on the shipped library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s18; mkdir -p $O
sha256sum lidar4d_amd/liblidar4d_hip.so | cut -c1-16 | tee $O/lib_sha.txt
for W in c5 c2; do
  B="python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt_$W -o kt -- $B > $GRAFT_REPO_ROOT/$O/kt_$W.log 2>&1 )
  python tools/rocpd_stats.py $(find $O/kt_$W -name "*.db" | head -1) 45 > $O/kernel_stats_$W.txt 2>&1
  rm -rf $O/kt_$W
  head -4 $O/kernel_stats_$W.txt | cut -c1-150
done
