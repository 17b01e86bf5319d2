#!/bin/bash
# round 3, run E: parity (fused prep, single histogram), graph diagnostics, A/B, counter calibration
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3e}
mkdir -p $O
python -m pytest tests/test_gpu_optim.py tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q --tb=short -k "not graphed" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 12
timeout 300 python tools/graph_probe.py 1024 > $O/probe_1k.log 2>&1; echo "probe rc=$?"; grep -E "eager|capture|replay|PROBE_OK|fault" $O/probe_1k.log | cut -c1-330
timeout 300 python -m pytest tests/test_gpu_optim.py -m gpu -q --tb=short -k "graphed" > $O/pytest_graph.log 2>&1; echo "pytest graph rc=$?"; grep -E "^E  |passed|failed" $O/pytest_graph.log | head -5 | cut -c1-300
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant-steps 0"
show() {
  python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("  unreadable:", e); sys.exit()
c = d["config"]
print("  ms/step %.2f  rays/s %.0f  skipped %s settle %s mode: %s" % (d["ms_per_step"], d["value"], c.get("skipped_steps_in_timed_region"), c.get("scaler_settling_steps_before_warmup"), c.get("step_mode", "")[:50]))
rk = d.get("roofline_kernels") or []
print("  sum of library + torch kernel ms in the profile pass: %.2f" % sum(r["ms_per_step"] for r in rk))
for r in rk[:20]:
    print("   %-60s %7.3f ms n=%.1f" % (r["kernel"][:60], r["ms_per_step"], r["launches_per_step"]))
PY
}
run() {  # name, env...
  name=$1; shift
  env "$@" $B --profile-steps 2 > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name rc=$?"
  show $O/bench_$name.json
}
run default L4D_STREAMS=0
run nofusedprep L4D_STREAMS=0 L4D_NO_FUSED_PREP=1
run streams2 L4D_STREAMS=2
for C in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | cut -d' ' -f1)
  ( cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/cal_$T -o c -- $GRAFT_REPO_ROOT/tools/ubench/calib_ubench > $GRAFT_REPO_ROOT/$O/calib_$T.log 2>&1 )
  python tools/rocpd_pmc.py $(find $O/cal_$T -name "*.db" | head -1) > $O/calib_$T.txt 2>&1
  grep -E "calib_|counter" $O/calib_$T.txt | head -40
  rm -rf $O/cal_$T
done
grep -E "^calib|bytes per" $O/calib_FETCH_SIZE.log
ls $O | head -40
