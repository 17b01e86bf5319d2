#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2c6
mkdir -p $O
python -m pytest tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 6
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0"
for V in "11 12" "12 12" "11 11" "11 10"; do
  set -- $V
  L4D_BS_SHIFT4=$1 L4D_BS_SHIFT2=$2 $B > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; echo "bench shift4=$1 shift2=$2 rc=$?"
  python - "$O/bench_$1_$2.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("  ms/step %.2f" % d["ms_per_step"], " ".join("%s=%.2f" % (r["kernel"][:22], r["ms_per_step"]) for r in d["roofline_kernels"] if r["kernel"].startswith("bin_pass")))
PY
done
