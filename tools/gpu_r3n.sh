#!/bin/bash
# round 3, run N: flow outputs in fp32 (gradient arrives unsaturated) -> loss scale; graph mode with side streams off, C3 and 1,024 rays
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3n}
mkdir -p $O
timeout 300 python tools/scale_probe.py 65536 110 > $O/scale_c3.log 2>&1; echo "scale c3 rc=$?"; grep -E "^step|final" $O/scale_c3.log | tail -n 12 | cut -c1-330
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_model.py tests/test_next_rows.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error|rc=" $O/pytest.log | tail -n 6
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["config"]
    print("  %s: ms/step %.2f  rays/s %.0f  settle %s skipped %s/%s scale %s mode %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], c.get("scaler_settling_steps_before_warmup"),
          c.get("skipped_steps_in_timed_region"), c.get("skipped_steps_in_warmup"), c.get("loss_scale_after_timed_region"), c.get("step_mode")))
except Exception as e:
    print("  %s unreadable: %r" % (sys.argv[1], e))
PY
}
B="python bench.py --no-cpu-baseline --variant-steps 0 --profile-steps 0"
$B --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; show $O/bench_default.json
$B --steps 20 --warmup 3 --graph > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench graph rc=$?"; show $O/bench_graph.json
$B --workload c3-1k --steps 60 --warmup 5 > $O/bench_1k.json 2> $O/bench_1k.err; echo "1k rc=$?"; show $O/bench_1k.json
$B --workload c3-1k --steps 60 --warmup 5 --graph > $O/bench_1k_graph.json 2> $O/bench_1k_graph.err; echo "1k graph rc=$?"; show $O/bench_1k_graph.json
