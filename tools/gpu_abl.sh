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
print("  ms/step %.2f" % d["ms_per_step"], " ".join("%s=%.2f" % (r["kernel"][:22], r["ms_per_step"]) for r in d["roofline_kernels"] if r["kernel"].startswith("bin_pass")))
PY
}
run default X=1
run b512_s10 L4D_LIB=$PWD/tools/abl/lib_bins512.so L4D_BS_SHIFT4=10
run b512_s10_s11 L4D_LIB=$PWD/tools/abl/lib_bins512.so L4D_BS_SHIFT4=10 L4D_BS_SHIFT2=11
run b512_s11 L4D_LIB=$PWD/tools/abl/lib_bins512.so L4D_BS_SHIFT4=11
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
