#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2c3
mkdir -p $O
python -m pytest tests -m gpu -q -s -rfE --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant-steps 0"
$B > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for V in bsu2 bsu4 bsg16u2; do
  L4D_LIB=$PWD/tools/abl/lib_$V.so $B > $O/bench_$V.json 2> $O/bench_$V.err; echo "bench $V rc=$?"
done
python - <<'PY'
import json
for f in ("bench", "bench_bsu2", "bench_bsu4", "bench_bsg16u2"):
    try:
        d = json.load(open(f"gpurun_out/r2c3/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "ms/step %.2f" % d["ms_per_step"], "rays/s %.0f" % d["value"])
    for r in d["roofline_kernels"][:(30 if f == "bench" else 40)]:
        if f == "bench" or r["kernel"].startswith("bin_pass"):
            print("   %-52s %7.3f ms n=%.1f frac=%s" % (r["kernel"][:52], r["ms_per_step"], r["launches_per_step"], r.get("frac", "-")))
PY
python tools/diag_frame50.py 2>&1 | grep "^frame" | tee $O/diag.txt
