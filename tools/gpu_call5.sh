#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2c5
mkdir -p $O
python -m pytest tests -m gpu -q -s -rfE --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --trained-steps 0"
$B > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2c5/bench.json"))
print("ms/step %.2f" % d["ms_per_step"], "rays/s %.0f" % d["value"], "three-loss", d["variants"]["three_loss_step"]["ms_per_step"] if d.get("variants") else None)
print("hash_encoder", json.dumps({k: v for k, v in d["hash_encoder"].items() if k.startswith("L")}))
for r in d["roofline_kernels"][:24]:
    print("   %-52s %7.3f ms n=%.1f frac=%s" % (r["kernel"][:52], r["ms_per_step"], r["launches_per_step"], r.get("frac", "-")))
PY
