#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2c2
mkdir -p $O
python -m pytest tests -m gpu -q -s -rfE --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant-steps 0"
$B > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
L4D_LIB=$PWD/tools/abl/lib_encw3.so $B > $O/bench_encw3.json 2> $O/bench_encw3.err; echo "bench encw3 rc=$?"
python - <<'PY'
import json
for f in ("bench", "bench_encw3"):
    try:
        d = json.load(open(f"gpurun_out/r2c2/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "ms/step %.2f" % d["ms_per_step"], "rays/s %.0f" % d["value"])
    for r in d["roofline_kernels"][:26]:
        print("   %-44s %7.3f ms n=%.1f frac=%s" % (r["kernel"][:44], r["ms_per_step"], r["launches_per_step"], r.get("frac", "-")))
PY
