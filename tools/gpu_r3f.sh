#!/bin/bash
# round 3, run F: parity incl. the tcnn-fp16-gradient yardstick, pass-2 bin order A/B, prep variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3f}
mkdir -p $O
python -m pytest tests/test_gpu_optim.py tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q --tb=short -s -k "not graphed" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
grep -A14 "gradient error against the exact" $O/pytest.log | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_optim.py -m gpu -q --tb=short -k "graphed" > $O/pytest_graph.log 2>&1; echo "pytest graph rc=$?"; grep -E "^E  |passed|failed" $O/pytest_graph.log | head -5 | cut -c1-300
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant-steps 0"
show() {
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("  unreadable:", e); sys.exit()
c = d["config"]
rk = d.get("roofline_kernels") or []
print("  ms/step %.2f  rays/s %.0f  skipped %s settle %s | kernels in the profile pass %.2f ms" % (d["ms_per_step"], d["value"], c.get("skipped_steps_in_timed_region"), c.get("scaler_settling_steps_before_warmup"), sum(r["ms_per_step"] for r in rk)))
for r in rk[:14]:
    print("   %-60s %7.3f ms n=%.1f" % (r["kernel"][:60], r["ms_per_step"], r["launches_per_step"]))
PY
}
run() {  # name, env...
  name=$1; shift
  env "$@" $B --profile-steps 2 > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"
  show $O/bench_$name.json
}
run default L4D_STREAMS=0
run noxcdbins L4D_STREAMS=0 L4D_LIB=$PWD/tools/abl/lib_noxcdbins.so
run streams2 L4D_STREAMS=2
run prepside L4D_STREAMS=2 L4D_NO_FUSED_PREP=1 L4D_PREP_SIDE=1
run nofused_s2 L4D_STREAMS=2 L4D_NO_FUSED_PREP=1
ls $O | head -40
