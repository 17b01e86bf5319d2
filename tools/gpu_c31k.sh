cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r4i
for v in 2048 8192; do
L4D_BWD_CHUNK_MIN=$v L4D_BENCH_DETAIL=$PWD/gpurun_out/r4i/c31k_$v.json python bench.py --workload c3-1k --steps 40 --warmup 5 --no-cpu-baseline --variant-steps 0 --profile-steps 2 > gpurun_out/r4i/line_$v.json 2>gpurun_out/r4i/err_$v.txt
python - <<PY
import json
d=json.load(open("gpurun_out/r4i/c31k_$v.json")); print("chunk_min $v: %.3f ms/step" % d["ms_per_step"])
for r in d["roofline_kernels"][:8]: print("   %-50s %.3f" % (r["kernel"][:50], r["ms_per_step"]))
PY
done
python -m pytest tests/test_gpu_c3_parity.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -2
