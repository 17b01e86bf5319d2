#!/bin/bash
# round 3, run L: loss-scale diagnosis (which gradients go non-finite) + graph fault split between the flow loss and the chamfer loss
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3l}
mkdir -p $O
timeout 300 python tools/scale_probe.py 65536 60 > $O/scale_c3.log 2>&1; echo "scale c3 rc=$?"; tail -n 25 $O/scale_c3.log | cut -c1-400
PROBE_FLOW=0 timeout 300 python tools/scale_probe.py 65536 40 > $O/scale_c3_noflow.log 2>&1; echo "scale c3 noflow rc=$?"; tail -n 12 $O/scale_c3_noflow.log | cut -c1-400
probe() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python tools/graph_probe.py 4096 > $O/probe_$name.log 2>&1; echo "probe $name rc=$? $(grep -E 'PROBE_OK|fault' $O/probe_$name.log | tail -1 | cut -c1-160)"
}
probe flowonly PROBE_CHAMFER=0
probe chamferonly PROBE_FLOW=0
