#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/grid; mkdir -p $O
for G in 512 256 384; do
  L4D_MLP_BWD_GRID=$G python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 0 > $O/b$G.json 2> $O/b$G.err
  python - $O/b$G.json $G <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("grid", sys.argv[2], "ms/step %.2f" % d["ms_per_step"], " ".join("%s=%.3f" % (r["kernel"][:26], r["ms_per_step"]) for r in d["roofline_kernels"] if r["kernel"].startswith("mlp_bwd")))
PY
done
