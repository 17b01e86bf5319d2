#!/bin/bash
# round-5 session 5: level-major pre-passes for the xy stack and the flow grid
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
timeout 600 bash tools/gpu_ab.sh s5 none all noxy:L4D_ENC_XY_SPLIT=0 noflowlv:L4D_FLOW_LEVELS=0 neither:L4D_ENC_XY_SPLIT=0,L4D_FLOW_LEVELS=0
env L4D_BENCH_DETAIL=$PWD/$O/tr_all_detail.json python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 8 --trained-steps 200 --profile-steps 0 > $O/tr_all.json 2> $O/tr_all.err
python - $O/tr_all.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], d.get("variants"))
PY
timeout 900 python -m pytest tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_properties.py tests/test_gpu_glue.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
