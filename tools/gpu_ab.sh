#!/bin/bash
# One parameterised GPU session (replaces the per-run scripts of round 3):
#   bash tools/gpu_ab.sh <tag> [tests=none|quick|full] [VARIANT ...]
# VARIANT = name              -> bench with L4D_LIB=tools/abl/lib_<name>.so (tools/build_abl.sh)
#         = name:ENV=V,ENV=V  -> bench with the in-tree library (or lib_<name>.so if it exists) under those env settings
#           (LIB=<other> picks tools/abl/lib_<other>.so; ARGS=--flag+value adds bench flags, '+' for blanks)
# Every bench is the short form (8 timed steps, per-kernel pass, no CPU baseline / variants); the per-kernel table goes to
# gpurun_out/<tag>/<name>.txt, the compact line to <name>.json, the side file to <name>_detail.json.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-ab}; TESTS=${2:-quick}; shift 2
O=gpurun_out/$TAG
mkdir -p $O
sha256sum lidar4d_amd/liblidar4d_hip.so | cut -c1-16 > $O/lib_sha.txt
if [ "$TESTS" == "quick" ]; then
  timeout 600 python -m pytest tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
  echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|Error|rc=" $O/pytest.log | tail -n 6
elif [ "$TESTS" == "full" ]; then
  timeout 900 python -m pytest tests -m gpu -q -rfE --tb=short > $O/pytest.log 2>&1
  echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|FAILED|Error|rc=" $O/pytest.log | tail -n 8
fi
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0"
run() {  # name, env assignments...
  name=$1; shift
  extra=""
  for a in "$@"; do [[ "$a" == ARGS=* ]] && extra=$(echo "${a#ARGS=}" | tr '+' ' '); done
  env L4D_BENCH_DETAIL=$PWD/$O/${name}_detail.json "$@" $B $extra > $O/$name.json 2> $O/$name.err; echo "bench $name rc=$? ($*)"
  python - "$O/$name.json" "$O/${name}_detail.json" > $O/$name.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); det = json.load(open(sys.argv[2]))
print("ms/step %.3f  line_bytes %d  skipped %s" % (d["ms_per_step"], len(open(sys.argv[1]).read()), d["config"]["skipped_steps_in_timed_region"]))
for r in det["roofline_kernels"][:28]:
    print("   %-56s %7.3f ms n=%.1f frac=%s" % (r["kernel"][:56], r["ms_per_step"], r["launches_per_step"], r.get("frac", "-")))
PY
  head -1 $O/$name.txt
}
for V in "$@"; do
  name=${V%%:*}; envs=""
  [[ "$V" == *:* ]] && envs=$(echo "${V#*:}" | tr ',' ' ')
  lib=""; [ -f tools/abl/lib_$name.so ] && lib="L4D_LIB=$PWD/tools/abl/lib_$name.so"
  for a in $envs; do [[ "$a" == LIB=* ]] && lib="L4D_LIB=$PWD/tools/abl/lib_${a#LIB=}.so"; done
  run $name X=1 $lib $envs
done
python - $O "$@" <<'PY'
import json, sys, os
O = sys.argv[1]; names = [v.split(":")[0] for v in sys.argv[2:]]
tabs = {}
for n in names:
    try:
        det = json.load(open(os.path.join(O, n + "_detail.json")))
        tabs[n] = ({r["kernel"]: r["ms_per_step"] for r in det["roofline_kernels"]}, det["ms_per_step"])
    except Exception as e:
        print(n, "unreadable", e)
if tabs:
    first = names[0] if names[0] in tabs else list(tabs)[0]
    keys = [k for k, v in sorted(tabs[first][0].items(), key=lambda kv: -kv[1])][:24]
    print("%-50s" % "kernel" + "".join("%10s" % n[:9] for n in tabs))
    print("%-50s" % "STEP (timed region)" + "".join("%10.2f" % tabs[n][1] for n in tabs))
    for k in keys:
        print("%-50s" % k[:50] + "".join("%10.3f" % tabs[n][0].get(k, float("nan")) for n in tabs))
PY
rm -f $O/*.err.empty
