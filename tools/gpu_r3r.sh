#!/bin/bash
# round 3, run R: the step's value-carrying reductions as HIP launches (no torch multi-block reduce, hence no memset node):
# do single-chain replays now equal eager?  graph mode of bench.py with side streams off
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3r}
mkdir -p $O
timeout 200 python tools/graph_diff.py 4096 12 > $O/diff_s0.log 2>&1; echo "diff s0 rc=$? $(grep GRAPH_DIFF $O/diff_s0.log)"; grep -E "^replay  [15]" $O/diff_s0.log | cut -c1-260
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_ops.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error|rc=" $O/pytest.log | tail -n 6
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
$B --workload c3-1k --steps 100 --warmup 5 --graph > $O/bench_1k_graph.json 2> $O/bench_1k_graph.err; echo "1k graph rc=$?"; show $O/bench_1k_graph.json
$B --workload c3-1k --steps 100 --warmup 5 > $O/bench_1k.json 2> $O/bench_1k.err; echo "1k eager rc=$?"; show $O/bench_1k.json
$B --steps 20 --warmup 3 --graph > $O/bench_graph.json 2> $O/bench_graph.err; echo "c3 graph rc=$?"; show $O/bench_graph.json
$B --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "c3 eager rc=$?"; show $O/bench_default.json
