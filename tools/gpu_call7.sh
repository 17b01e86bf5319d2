#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2c7
mkdir -p $O
python -m pytest tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 6
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0"
for V in default bst256; do
  if [ $V = default ]; then unset L4D_LIB; else export L4D_LIB=$PWD/tools/abl/lib_$V.so; fi
  $B > $O/bench_$V.json 2> $O/bench_$V.err; echo "bench $V rc=$?"
  python - "$O/bench_$V.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("  ms/step %.2f" % d["ms_per_step"])
for r in d["roofline_kernels"][:16]:
    print("   %-52s %7.3f ms n=%.1f" % (r["kernel"][:52], r["ms_per_step"], r["launches_per_step"]))
PY
done
