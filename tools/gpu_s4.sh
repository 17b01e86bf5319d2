#!/bin/bash
# round-5 session 4: x-neighbour pair loads (PAIRLD) in the level-major pre-pass and inside the encode kernel; early and trained state
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --tb=short -k "hashgrid" > $O/pytest_hash.log 2>&1; echo "pytest hashgrid rc=$?"; tail -n 2 $O/pytest_hash.log
timeout 600 bash tools/gpu_ab.sh s4 none split_pair nosplit_pair:L4D_ENC_HS_SPLIT=0 base:L4D_ENC_HS_SPLIT=0,L4D_HS_PAIRLD=0 split_nopair:L4D_HS_PAIRLD=0
# trained state (variants.trained_state) of the two candidates
for v in "split_pair X=1" "nosplit_pair L4D_ENC_HS_SPLIT=0" "base L4D_ENC_HS_SPLIT=0 L4D_HS_PAIRLD=0"; do
  set -- $v; name=$1; shift
  env "$@" L4D_BENCH_DETAIL=$PWD/$O/tr_${name}_detail.json python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 8 --trained-steps 200 --profile-steps 0 > $O/tr_$name.json 2> $O/tr_$name.err
  python - $O/tr_$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], d.get("variants"), d.get("hash_encoder"))
PY
done
timeout 400 python -m pytest tests/test_gpu_c3_parity.py tests/test_gpu_model.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
