#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/p1k
mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --workload c3-1k --steps 20 --warmup 3 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o kt -- $B > $GRAFT_REPO_ROOT/$O/kt.log 2>&1 )
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 45 > $O/kernel_stats_1k.txt 2>&1
rm -rf $O/kt
head -50 $O/kernel_stats_1k.txt | cut -c1-150
