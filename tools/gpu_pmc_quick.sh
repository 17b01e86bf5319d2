#!/bin/bash
# A few rocprofv3 --pmc passes of the short bench command (no tests, no other workloads): per-kernel counter sums in
# gpurun_out/<tag>/pmc_<name>.txt.   usage: bash tools/gpu_pmc_quick.sh <tag> <name>:<COUNTER,COUNTER,...> [...]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-pmcq}; shift
O=gpurun_out/$TAG
mkdir -p $O
S="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 12 --settle-max 0 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
for spec in "$@"; do
  name=${spec%%:*}; ctrs=$(echo "${spec#*:}" | tr ',' ' ')
  ( cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$name -o p -- $S > $GRAFT_REPO_ROOT/$O/pmc_$name.log 2>&1 )
  python tools/rocpd_pmc.py $(find $O/pmc_$name -name "*.db" | head -1) --json $O/pmc_$name.json > $O/pmc_$name.txt 2>&1
  echo "pmc $name: $(wc -l < $O/pmc_$name.txt) lines"
  rm -rf $O/pmc_$name $O/pmc_$name.log
done
