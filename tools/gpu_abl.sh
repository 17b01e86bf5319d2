#!/bin/bash
# A/B of tuning builds (tools/build_abl.sh) against the in-tree library: quick parity tests on the default, then one short bench each
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-abl}
shift
mkdir -p $O
python -m pytest tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 6
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0"
run() {  # name, env...
  name=$1; shift
  env "$@" $B > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"
  python - "$O/bench_$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("  ms/step %.2f  hash_encoder %s" % (d["ms_per_step"], json.dumps(d.get("hash_encoder"))[:300]))
for r in d["roofline_kernels"][:22]:
    print("   %-52s %7.3f ms n=%.1f frac=%s" % (r["kernel"][:52], r["ms_per_step"], r["launches_per_step"], r.get("frac", "-")))
PY
}
run default X=1
for V in "$@"; do run $V L4D_LIB=$PWD/tools/abl/lib_$V.so; done
