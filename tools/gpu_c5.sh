#!/bin/bash
# inference workload with / without the captured staged chunk + the staged-render property test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-c5g}
mkdir -p $O
python -m pytest tests/test_gpu_properties.py -m gpu -q -x --tb=short -k "render_invariants" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 15 $O/pytest.log
for V in graph nograph; do
  F=""; [ $V = graph ] && F="--graph-staged"
  python bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 $F > $O/bench_c5_$V.json 2> $O/bench_c5_$V.err; echo "c5 $V rc=$?"
  python -c "import json;d=json.load(open('$O/bench_c5_$V.json'));print('$V %.2f ms/frame %.0f rays/s'%(d['ms_per_step'],d['value']))" || tail -n 12 $O/bench_c5_$V.err
done
