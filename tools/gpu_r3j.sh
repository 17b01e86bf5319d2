#!/bin/bash
# round 3, run J: pass 2 with compile-time bin size; quick parity + bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3j}
mkdir -p $O
python -m pytest tests/test_gpu_properties.py tests/test_gpu_ops.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 6
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant-steps 0 --profile-steps 2 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - $O/bench_default.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rk = d.get("roofline_kernels") or []
print("  ms/step %.2f  rays/s %.0f | kernels in the profile pass %.2f ms" % (d["ms_per_step"], d["value"], sum(r["ms_per_step"] for r in rk)))
for r in rk[:16]:
    print("   %-60s %7.3f ms n=%.1f" % (r["kernel"][:60], r["ms_per_step"], r["launches_per_step"]))
PY
