#!/bin/bash
# round-5 session 2: static grid through the level-major pre-pass (block order chip-wide vs XCD-pinned) against the in-kernel gathers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
timeout 500 bash tools/gpu_ab.sh s2 none default nosplit:L4D_ENC_HS_SPLIT=0 xcd:L4D_HG_ORDER=0
for n in default nosplit xcd; do python - $O/$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1], d.get("hash_encoder"), d["roofline"].get("kernel"), d["roofline"].get("frac"))
PY
done
timeout 600 python -m pytest tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 5 $O/pytest.log
