#!/bin/bash
# round 3, run C: parity (everything but the graph test), graph-capture probes in separate processes, benches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3c}
mkdir -p $O
python -m pytest tests/test_gpu_optim.py tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q --tb=short -k "not graphed" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
probe() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python tools/graph_probe.py 4096 > $O/probe_$name.log 2>&1; echo "probe $name rc=$? $(grep -E 'PROBE_OK|eager ' $O/probe_$name.log | tail -1 | cut -c1-200)"
}
probe in_s0 L4D_STREAMS=0
probe out_s0 L4D_STREAMS=0 L4D_GRAPH_BATCH=outside
probe in_s2 L4D_STREAMS=2
probe in_s2g L4D_STREAMS=2 L4D_GRAPH_STREAMS=1
python -m pytest tests/test_gpu_optim.py -m gpu -q --tb=short -k "graphed" > $O/pytest_graph.log 2>&1; echo "pytest graph rc=$?"; tail -3 $O/pytest_graph.log
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0"
show() {
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("  unreadable:", e); sys.exit()
print("  ms/step %.2f  rays/s %.0f  mode: %s" % (d["ms_per_step"], d["value"], d["config"].get("step_mode", "")[:70]))
for r in (d.get("roofline_kernels") or [])[:24]:
    print("   %-60s %7.3f ms n=%.1f" % (r["kernel"][:60], r["ms_per_step"], r["launches_per_step"]))
PY
}
run() {  # name, env...
  name=$1; shift
  env "$@" $B --no-graph --profile-steps 2 > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"
  show $O/bench_$name.json
}
run default L4D_STREAMS=0
run norecomp L4D_STREAMS=0 L4D_ATTR_RECOMP=0
run bs1024 L4D_STREAMS=0 L4D_LIB=$PWD/tools/abl/lib_bs1024.so
run bs1024g4 L4D_STREAMS=0 L4D_LIB=$PWD/tools/abl/lib_bs1024g4.so
$B --profile-steps 0 > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench graph rc=$?"; show $O/bench_graph.json; tail -2 $O/bench_graph.err
python bench.py --workload c3-1k --steps 30 --warmup 5 --no-cpu-baseline --variant-steps 0 --profile-steps 0 > $O/bench_1k_graph.json 2> $O/bench_1k_graph.err; echo "1k graph rc=$?"; show $O/bench_1k_graph.json
ls $O | head -50
