#!/bin/bash
# round 3, run M: chamfer workspace filled by a kernel (no memset node) -> graph replay with side streams off?; flow adjoint with
# device-side fp16 normalisation -> loss scale; settle-then-restore bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3m}
mkdir -p $O
probe() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python tools/graph_probe.py 4096 > $O/probe_$name.log 2>&1; echo "probe $name rc=$? $(grep -E 'PROBE_OK|fault' $O/probe_$name.log | tail -1 | cut -c1-160)"
}
probe plain X=1
if ! grep -q PROBE_OK $O/probe_plain.log; then probe seppool L4D_GRAPH_POOL=separate; fi
timeout 300 python tools/scale_probe.py 65536 70 > $O/scale_c3.log 2>&1; echo "scale c3 rc=$?"; grep -E "^step|final" $O/scale_c3.log | tail -n 14 | cut -c1-330
timeout 600 python -m pytest tests/test_gpu_optim.py tests/test_gpu_model.py -m gpu -q --tb=short -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error|rc=" $O/pytest.log | tail -n 6
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --variant-steps 0 --profile-steps 0 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - $O/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
print("  ms/step %.2f  rays/s %.0f  settle %s skipped %s/%s scale %s" % (d["ms_per_step"], d["value"], c.get("scaler_settling_steps_before_warmup"),
      c.get("skipped_steps_in_timed_region"), c.get("skipped_steps_in_warmup"), c.get("loss_scale_after_timed_region")))
PY
