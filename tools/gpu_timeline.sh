#!/bin/bash
# Kernel trace of a short bench run, then the timeline of one steady-state step (tools/rocpd_timeline.py).
#   usage: bash tools/gpu_timeline.sh <tag> [extra bench flags]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-tl}; shift
# (L4D_LIB in the environment picks another library)
O=gpurun_out/$TAG
mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --variant-steps 0 --profile-steps 0 --trained-steps 0 $*"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/kt -o kt -- $B > $GRAFT_REPO_ROOT/$O/kt.log 2>&1 )
DB=$(find $O/kt -name "*.db" | head -1)
echo "db: $DB"
python tools/rocpd_timeline.py $DB 3 3 > $O/timeline.txt 2>&1
tail -n 4 $O/timeline.txt
tail -n 2 $O/kt.log
