#!/bin/bash
# round-5 session 6: density network as the encode kernel's epilogue; recomputed activations in its backward
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_properties.py -m gpu -q -x --tb=short -k "epilogue or level_major_flow" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -n 3 $O/pytest_new.log
timeout 600 bash tools/gpu_ab.sh s6 none fused nosigma:L4D_ENC_SIGMA=0 recomp:L4D_MLP_RECOMP_SIGMA=1
env L4D_BENCH_DETAIL=$PWD/$O/tr_fused_detail.json python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 8 --trained-steps 200 --profile-steps 0 > $O/tr_fused.json 2> $O/tr_fused.err
python - $O/tr_fused.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], d.get("variants"), d["config"].get("trained_state"))
PY
timeout 900 python -m pytest tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_properties.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
