#!/bin/bash
# round 3, run Q: which nodes make single-chain replays differ from eager?  torch's zero fills as kernels / separate prep kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3q}
mkdir -p $O
run() { name=$1; shift; env "$@" timeout 200 python tools/graph_diff.py 4096 6 > $O/diff_$name.log 2>&1; echo "$name rc=$? $(grep GRAPH_DIFF $O/diff_$name.log)"; grep -E "^replay  [15]" $O/diff_$name.log | cut -c1-260; }
run patchzeros PATCH_ZEROS=1
run nofusedprep L4D_NO_FUSED_PREP=1
run both PATCH_ZEROS=1 L4D_NO_FUSED_PREP=1
run noflowcham PROBE_FLOW=0 PROBE_CHAMFER=0
