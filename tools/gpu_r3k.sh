#!/bin/bash
# round 3, run K: graph-capture bisection (side streams off) + validation of the pass-1 / index changes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3k}
mkdir -p $O
probe() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python tools/graph_probe.py 4096 > $O/probe_$name.log 2>&1; echo "probe $name rc=$? $(grep -E 'PROBE_OK|fault' $O/probe_$name.log | tail -1 | cut -c1-160)"
}
probe plain X=1
probe keepws L4D_GRAPH_KEEP_WS=1
probe nofused L4D_NO_FUSED_PREP=1
probe noflow PROBE_FLOW=0 PROBE_CHAMFER=0
probe streams2 L4D_STREAMS=2
probe batchout L4D_GRAPH_BATCH=outside
python -m pytest tests/test_gpu_properties.py tests/test_gpu_ops.py tests/test_gpu_c3_parity.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 6
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant-steps 0 --profile-steps 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - $O/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rk = d.get("roofline_kernels") or []
print("  ms/step %.2f  rays/s %.0f | kernels in the profile pass %.2f ms" % (d["ms_per_step"], d["value"], sum(r["ms_per_step"] for r in rk)))
for r in rk[:14]:
    print("   %-60s %7.3f ms n=%.1f" % (r["kernel"][:60], r["ms_per_step"], r["launches_per_step"]))
PY
