#!/bin/bash
# quick GPU check: the model / op / property parity tests + a short bench with the per-kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-quick}
mkdir -p $O
python -m pytest tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 6
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - "$O/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("  ms/step %.2f  rays/s %.0f" % (d["ms_per_step"], d["value"]))
for r in d["roofline_kernels"][:30]:
    print("   %-52s %7.3f ms n=%.1f frac=%s" % (r["kernel"][:52], r["ms_per_step"], r["launches_per_step"], r.get("frac", "-")))
PY
