#!/bin/bash
# round 3, run O: graph mode of bench.py with every memset / memcpy node of the library replaced by kernels (side streams off / on)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3o}
mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["config"]
    print("  %s: ms/step %.2f  rays/s %.0f  settle %s skipped %s/%s scale %s mode %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], c.get("scaler_settling_steps_before_warmup"),
          c.get("skipped_steps_in_timed_region"), c.get("skipped_steps_in_warmup"), c.get("loss_scale_after_timed_region"), c.get("step_mode")[:60]))
except Exception as e:
    print("  %s unreadable: %r" % (sys.argv[1], e))
PY
}
B="python bench.py --no-cpu-baseline --variant-steps 0 --profile-steps 0"
export L4D_BENCH_TRACE=1
$B --workload c3-1k --steps 60 --warmup 5 --graph > $O/bench_1k_graph.json 2> $O/bench_1k_graph.err; echo "1k graph rc=$? $(grep -E 'captured|settle' $O/bench_1k_graph.err | tail -1)"; show $O/bench_1k_graph.json
L4D_STREAMS=2 $B --workload c3-1k --steps 60 --warmup 5 --graph > $O/bench_1k_graph_s2.json 2> $O/bench_1k_graph_s2.err; echo "1k graph streams2 rc=$? $(grep -E 'captured|settle' $O/bench_1k_graph_s2.err | tail -1)"; show $O/bench_1k_graph_s2.json
L4D_STREAMS=2 $B --workload c3-1k --steps 60 --warmup 5 > $O/bench_1k_s2.json 2> $O/bench_1k_s2.err; echo "1k eager streams2 rc=$?"; show $O/bench_1k_s2.json
$B --steps 20 --warmup 3 --graph > $O/bench_graph.json 2> $O/bench_graph.err; echo "c3 graph rc=$? $(grep -E 'captured|settle' $O/bench_graph.err | tail -1)"; show $O/bench_graph.json
if ! [ -s $O/bench_graph.json ]; then
  L4D_STREAMS=2 $B --steps 20 --warmup 3 --graph > $O/bench_graph_s2.json 2> $O/bench_graph_s2.err; echo "c3 graph streams2 rc=$? $(grep -E 'captured|settle' $O/bench_graph_s2.err | tail -1)"; show $O/bench_graph_s2.json
fi
