#!/bin/bash
# round-5 session 10: the last level's launch assembles the rows (static grid stand-alone, flow grid)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_properties.py tests/test_gpu_ops.py -m gpu -q --tb=short -k "level_major or hashgrid or epilogue" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -n 4 $O/pytest_new.log
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 2 --trained-steps 0"
for v in "asm X=1" "noasm L4D_HG_ASSEMBLE=0" "base L4D_LIB=$PWD/tools/abl/lib_base.so"; do
  set -- $v; name=$1; shift
  env L4D_BENCH_DETAIL=$PWD/$O/${name}_detail.json "$@" $B > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $O/${name}_detail.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); det = json.load(open(sys.argv[2]))
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], d.get("hash_encoder"))
for r in det["roofline_kernels"][:26]:
    if "hashgrid" in r["kernel"] or "mlp_fwd_kernel<1" in r["kernel"]: print("   %-56s %7.3f ms" % (r["kernel"][:56], r["ms_per_step"]))
PY
done
