#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-abl}
mkdir -p $O
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0"
run() {  # name, env...
  name=$1; shift
  env "$@" $B > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"
  python - "$O/bench_$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
keep = ("planes_static", "planes_dyn", "field_bwd_prep", "dynhash_lds")
print("  ms/step %.2f" % d["ms_per_step"], " ".join("%s=%.2f" % (r["kernel"][:22], r["ms_per_step"]) for r in d["roofline_kernels"] if r["kernel"].startswith(keep)))
PY
}
run default X=1
for V in pdyn768; do run $V L4D_LIB=$PWD/tools/abl/lib_$V.so; done
