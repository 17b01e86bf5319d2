#!/bin/bash
# Round artefacts on the GPU box: full -m gpu test run, the default bench line, rocprofv3 kernel trace of the bench command,
# PMC passes (fabric traffic, MFMA counters) and the FETCH_SIZE calibration.  Summaries land in gpurun_out/<tag>/; copy the
# ones to keep into profiles/.   usage: bash tools/gpu_profile_round.sh <tag> [skip-tests]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r02}
O=gpurun_out/$TAG
mkdir -p $O
if [ "$2" != "skip-tests" ]; then
  python -m pytest tests -m gpu -q -s -rfE --tb=short > $O/pytest.log 2>&1
  echo "pytest rc=$?" >> $O/pytest.log
  grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 8
fi
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; head -c 600 $O/bench.json; echo
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o kt -- $B > $GRAFT_REPO_ROOT/$O/kt.log 2>&1 )
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 70 > $O/kernel_stats.txt 2>&1
S="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA"; do
  T=$(echo $C | cut -d' ' -f1)
  ( cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$T -o p -- $S > $GRAFT_REPO_ROOT/$O/pmc_$T.log 2>&1 )
  python tools/rocpd_pmc.py $(find $O/pmc_$T -name "*.db" | head -1) --json $O/pmc_$T.json > $O/pmc_$T.txt 2>&1
done
python tools/pmc_to_profiles.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES.json $O $TAG
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/$O/calib -o c -- python $GRAFT_REPO_ROOT/tools/gather_calib.py > $GRAFT_REPO_ROOT/$O/calib.log 2>&1 )
python tools/rocpd_pmc.py $(find $O/calib -name "*.db" | head -1) > $O/calib_pmc.txt 2>&1
grep -E "^width|^streaming" $O/calib.log >> $O/calib_pmc.txt
# the other workloads and the data-parallel code path (one forced rank: RCCL init, hook, async all-reduce, both transports)
python bench.py --workload c2 --steps 10 --no-cpu-baseline --variant-steps 0 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --profile-steps 0 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
D="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
L4D_FORCE_DIST=1 $D > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist1 rc=$?"
L4D_FORCE_DIST=1 L4D_GRAD_TRANSPORT=bf16 $D > $O/bench_dist1_bf16.json 2> $O/bench_dist1_bf16.err; echo "dist1 bf16 rc=$?"
python - <<PY
import json
for f in ("bench", "bench_c2", "bench_c5", "bench_dist1", "bench_dist1_bf16"):
    try:
        d = json.load(open("$O/%s.json" % f))
        print(f, "%.2f ms/step  %.0f rays/s  n_gpus %d" % (d["ms_per_step"], d["value"], d["n_gpus"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/calib $O/*.log
ls $O
