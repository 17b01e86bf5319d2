#!/bin/bash
# round-5 session 14: what the driver runs at round end, on the final library -- smoke(), then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s14; mkdir -p $O
sha256sum lidar4d_amd/liblidar4d_hip.so | cut -c1-16
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 4 $O/smoke.log
SECONDS=0
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall ${SECONDS}s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/s14/bench.json"))
print("ms/step %.3f value %.1f bytes %d skipped %s" % (d["ms_per_step"], d["value"], len(open("gpurun_out/s14/bench.json").read()), d["config"].get("skipped_steps_in_timed_region")))
PY
