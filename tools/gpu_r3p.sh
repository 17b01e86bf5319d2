#!/bin/bash
# round 3, run P: replayed step vs eager step, tensor by tensor, side streams off / on
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3p}
mkdir -p $O
timeout 300 python tools/graph_diff.py 4096 24 > $O/diff_s0.log 2>&1; echo "s0 rc=$?"; grep -vE "amdgpu.ids|Warning|warn" $O/diff_s0.log | tail -n 30 | cut -c1-420
L4D_STREAMS=2 timeout 300 python tools/graph_diff.py 4096 12 > $O/diff_s2.log 2>&1; echo "s2 rc=$?"; grep -vE "amdgpu.ids|Warning|warn" $O/diff_s2.log | tail -n 16 | cut -c1-420
