#!/bin/bash
# round 3, run H: LDS bank-conflict fixes (pass 2 value-major accumulators, rotated plane channel slots), yardstick test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3h}
mkdir -p $O
python -m pytest tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --tb=short -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
grep -A8 "HIP ran at total loss scale" $O/pytest.log | cut -c1-260
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant-steps 0"
show() {
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("  unreadable:", e); sys.exit()
rk = d.get("roofline_kernels") or []
print("  ms/step %.2f  rays/s %.0f | kernels in the profile pass %.2f ms" % (d["ms_per_step"], d["value"], sum(r["ms_per_step"] for r in rk)))
for r in rk[:18]:
    print("   %-60s %7.3f ms n=%.1f" % (r["kernel"][:60], r["ms_per_step"], r["launches_per_step"]))
PY
}
run() {  # name, env...
  name=$1; shift
  env "$@" $B --profile-steps 2 > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"
  show $O/bench_$name.json
}
run default L4D_STREAMS=0
run bsu1 L4D_STREAMS=0 L4D_LIB=$PWD/tools/abl/lib_bsu1.so
run streams2 L4D_STREAMS=2
ls $O | head -40
