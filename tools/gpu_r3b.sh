#!/bin/bash
# round 3, run B: parity with pair records + graphed step tests, A/B of the sorted-scatter variants, flow-net knobs, graph mode
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3b}
mkdir -p $O
python -m pytest tests/test_gpu_optim.py tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0"
show() {
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("  unreadable:", e); sys.exit()
print("  ms/step %.2f  rays/s %.0f  mode: %s" % (d["ms_per_step"], d["value"], d["config"].get("step_mode", "")[:60]))
want = ("bin_pass", "mlp_bwd_kernel<1, 2", "mlp_fwd_kernel<1, 2", "hashgrid_t_fwd")
for r in (d.get("roofline_kernels") or []):
    if any(w in r["kernel"] for w in want):
        print("   %-60s %7.3f ms n=%.1f" % (r["kernel"][:60], r["ms_per_step"], r["launches_per_step"]))
PY
}
run() {  # name, env...
  name=$1; shift
  env "$@" $B --no-graph --profile-steps 2 > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"
  show $O/bench_$name.json
}
run default L4D_STREAMS=0
for V in oldbs bs1024 bsg4 bsg16 bs1024g4; do run $V L4D_STREAMS=0 L4D_LIB=$PWD/tools/abl/lib_$V.so; done
run flowgrid2 L4D_STREAMS=0 L4D_MLP_BWD_GRID_NARROW=2
run flowgrid3 L4D_STREAMS=0 L4D_MLP_BWD_GRID_NARROW=3
run flowstore L4D_STREAMS=0 L4D_MLP_STORE_ACT=1
# graph mode vs eager, default streams
$B --profile-steps 0 > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench graph rc=$?"; show $O/bench_graph.json; tail -3 $O/bench_graph.err
$B --no-graph --profile-steps 0 > $O/bench_eager.json 2> $O/bench_eager.err; echo "bench eager rc=$?"; show $O/bench_eager.json
python bench.py --workload c3-1k --steps 30 --warmup 5 --no-cpu-baseline --variant-steps 0 --profile-steps 0 > $O/bench_1k_graph.json 2> $O/bench_1k_graph.err; echo "1k graph rc=$?"; show $O/bench_1k_graph.json
python bench.py --workload c3-1k --steps 30 --warmup 5 --no-cpu-baseline --variant-steps 0 --profile-steps 0 --no-graph > $O/bench_1k_eager.json 2> $O/bench_1k_eager.err; echo "1k eager rc=$?"; show $O/bench_1k_eager.json
ls $O | head -50
