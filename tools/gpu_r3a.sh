#!/bin/bash
# round 3, run A: parity subset under side streams, stream-mask A/B of the default step, TCC counters on the forward encode
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3a}
mkdir -p $O
python -m pytest tests/test_gpu_properties.py tests/test_gpu_c3_parity.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 6
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --variant-steps 0"
for M in 0 1 2 3; do
  PS=0; [ $M = 0 ] && PS=2
  L4D_STREAMS=$M $B --profile-steps $PS > $O/bench_s$M.json 2> $O/bench_s$M.err; echo "bench streams=$M rc=$?"
  python - "$O/bench_s$M.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("  ms/step %.2f  rays/s %.0f" % (d["ms_per_step"], d["value"]))
for r in (d.get("roofline_kernels") or [])[:34]:
    print("   %-60s %7.3f ms n=%.1f" % (r["kernel"][:60], r["ms_per_step"], r["launches_per_step"]))
PY
done
# counters available on this box
( cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1 )
grep -c . $O/counters_list.txt
S="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  T=$(echo $C | tr ' ' '+')
  ( cd /tmp && L4D_STREAMS=0 timeout 300 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$T -o p -- $S > $GRAFT_REPO_ROOT/$O/pmc_$T.log 2>&1 )
  python tools/rocpd_pmc.py $(find $O/pmc_$T -name "*.db" | head -1) --json $O/pmc_$T.json > $O/pmc_$T.txt 2>&1
  head -c 1500 $O/pmc_$T.txt; echo
  rm -rf $O/pmc_$T
done
ls $O
