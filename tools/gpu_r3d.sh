#!/bin/bash
# round 3, run D: parity, out-of-bounds hunt, graph test / probes, scatter variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3d}
mkdir -p $O
python -m pytest tests/test_gpu_optim.py tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q --tb=short -k "not graphed" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
for N in 1024 16384; do
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 L4D_TRACE=1 L4D_STREAMS=0 timeout 600 python tools/oob_probe.py $N > $O/oob_$N.log 2>&1; echo "oob $N rc=$? $(grep -c 'launch' $O/oob_$N.log) launches; $(grep OOB_PROBE $O/oob_$N.log | tail -1)"
  grep -B2 -A6 -i "fault\|error\|abort" $O/oob_$N.log | grep -v "amdgpu.ids" | head -20
  tail -n 400 $O/oob_$N.log > $O/oob_$N.tail; mv $O/oob_$N.tail $O/oob_$N.log
done
python -m pytest tests/test_gpu_optim.py -m gpu -q --tb=short -k "graphed" > $O/pytest_graph.log 2>&1; echo "pytest graph rc=$?"; tail -3 $O/pytest_graph.log
probe() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python tools/graph_probe.py 4096 > $O/probe_$name.log 2>&1; echo "probe $name rc=$? $(grep -E 'PROBE_OK|eager ' $O/probe_$name.log | tail -1 | cut -c1-200)"
}
probe default X=1
probe nostreams L4D_GRAPH_STREAMS=0
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0"
show() {
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("  unreadable:", e); sys.exit()
print("  ms/step %.2f  rays/s %.0f  mode: %s" % (d["ms_per_step"], d["value"], d["config"].get("step_mode", "")[:70]))
for r in (d.get("roofline_kernels") or [])[:16]:
    print("   %-60s %7.3f ms n=%.1f" % (r["kernel"][:60], r["ms_per_step"], r["launches_per_step"]))
PY
}
run() {  # name, env...
  name=$1; shift
  env "$@" $B --profile-steps 2 > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"
  show $O/bench_$name.json
}
run default L4D_STREAMS=0
run shift12 L4D_STREAMS=0 L4D_BS_SHIFT4=12
run shift12_13 L4D_STREAMS=0 L4D_BS_SHIFT4=12 L4D_BS_SHIFT2=13
run bs1024 L4D_STREAMS=0 L4D_LIB=$PWD/tools/abl/lib_bs1024.so
run bs1024s12 L4D_STREAMS=0 L4D_BS_SHIFT4=12 L4D_LIB=$PWD/tools/abl/lib_bs1024.so
run streams2 L4D_STREAMS=2
$B --graph --profile-steps 0 > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench graph rc=$?"; show $O/bench_graph.json; tail -2 $O/bench_graph.err | cut -c1-300
python bench.py --workload c3-1k --steps 30 --warmup 5 --no-cpu-baseline --variant-steps 0 --profile-steps 0 --graph > $O/bench_1k_graph.json 2> $O/bench_1k_graph.err; echo "1k graph rc=$?"; show $O/bench_1k_graph.json
python bench.py --workload c3-1k --steps 30 --warmup 5 --no-cpu-baseline --variant-steps 0 --profile-steps 0 > $O/bench_1k_eager.json 2> $O/bench_1k_eager.err; echo "1k eager rc=$?"; show $O/bench_1k_eager.json
ls $O | head -50
