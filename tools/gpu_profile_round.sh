#!/bin/bash
# Round artefacts on the GPU box: full -m gpu test run, rocprofv3 kernel trace of the bench command, PMC passes (memory-side
# request counters, L2 hit / miss, MFMA, SQ / LDS), the counter calibration, the default bench line (with the counter files of THIS
# build attached) and the other workloads.  Summaries land in gpurun_out/<tag>/; copy the ones to keep into profiles/.
#   usage: bash tools/gpu_profile_round.sh <tag> [skip-tests|subset]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r03}
O=gpurun_out/$TAG
mkdir -p $O
if [ "$2" == "subset" ]; then
  python -m pytest tests/test_gpu_ops.py tests/test_gpu_c3_parity.py tests/test_gpu_optim.py -m gpu -q --tb=short > $O/pytest_subset.log 2>&1
  echo "pytest rc=$?" >> $O/pytest_subset.log
  grep -E "passed|failed|FAILED|Error" $O/pytest_subset.log | tail -n 8
elif [ "$2" != "skip-tests" ]; then
  python -m pytest tests -m gpu -q -s -rfE --tb=short > $O/pytest.log 2>&1
  echo "pytest rc=$?" >> $O/pytest.log
  grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -n 8
fi
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o kt -- $B > $GRAFT_REPO_ROOT/$O/kt.log 2>&1 )
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 70 > $O/kernel_stats.txt 2>&1
head -5 $O/kernel_stats.txt | cut -c1-160
S="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 12 --settle-max 0 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
pmc() {  # name, counters...
  name=$1; shift
  ( cd /tmp && timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$name -o p -- $S > $GRAFT_REPO_ROOT/$O/pmc_$name.log 2>&1 )
  python tools/rocpd_pmc.py $(find $O/pmc_$name -name "*.db" | head -1) --json $O/pmc_$name.json > $O/pmc_$name.txt 2>&1
  echo "pmc $name: $(wc -l < $O/pmc_$name.txt) lines"
  rm -rf $O/pmc_$name
}
pmc rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pmc wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum
pmc l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_WAVES
python tools/pmc_summary.py $O/pmc_rd.json $O/pmc_wr.json $O/pmc_l2.json $O/pmc_mfma.json $O $TAG lidar4d_amd/liblidar4d_hip.so
cp $O/hbm_traffic_$TAG.json $O/${TAG}_mfma_pmc.json profiles/   # the bench lines below attach them (same build: sha256 checked)
for C in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum"; do
  T=$(echo $C | cut -d' ' -f1)
  ( cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/cal_$T -o c -- $GRAFT_REPO_ROOT/tools/ubench/calib_ubench > $GRAFT_REPO_ROOT/$O/calib_$T.log 2>&1 )
  python tools/rocpd_pmc.py $(find $O/cal_$T -name "*.db" | head -1) > $O/calib_$T.txt 2>&1
  rm -rf $O/cal_$T
done
{ grep -E "^calib|bytes per" $O/calib_FETCH_SIZE.log; grep -E "calib_|counter" $O/calib_FETCH_SIZE.txt $O/calib_TCC_EA0_RDREQ_sum.txt; } > $O/${TAG}_counter_calibration.txt
L4D_BENCH_DETAIL=$PWD/$O/bench_detail.json python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; head -c 400 $O/bench.json; echo
# the other workloads and the data-parallel code path (one forced rank: RCCL init, hook, async all-reduce, both transports)
L4D_BENCH_DETAIL=$PWD/$O/bench_graph_detail.json python bench.py --graph --steps 20 --warmup 5 --no-cpu-baseline --variant-steps 0 --profile-steps 0 > $O/bench_graph.json 2> $O/bench_graph.err; echo "graph rc=$?"
L4D_BENCH_DETAIL=$PWD/$O/bench_c2_detail.json python bench.py --workload c2 --steps 10 --no-cpu-baseline --variant-steps 0 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
L4D_BENCH_DETAIL=$PWD/$O/bench_c5_detail.json python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --profile-steps 0 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
L4D_BENCH_DETAIL=$PWD/$O/bench_c3_1k_detail.json python bench.py --workload c3-1k --steps 40 --warmup 5 --no-cpu-baseline --variant-steps 0 --profile-steps 2 > $O/bench_c3_1k.json 2> $O/bench_c3_1k.err; echo "c3-1k rc=$?"
L4D_BENCH_DETAIL=$PWD/$O/bench_c3_1k_graph_detail.json python bench.py --workload c3-1k --graph --steps 40 --warmup 5 --no-cpu-baseline --variant-steps 0 --profile-steps 0 > $O/bench_c3_1k_graph.json 2> $O/bench_c3_1k_graph.err; echo "c3-1k graph rc=$?"
export L4D_BENCH_DETAIL=$PWD/$O/bench_dist_detail.json
D="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
$D --force-dist > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist1 rc=$?"
$D --force-dist --grad-transport bf16 > $O/bench_dist1_bf16.json 2> $O/bench_dist1_bf16.err; echo "dist1 bf16 rc=$?"
python - <<PY
import json
for f in ("bench", "bench_graph", "bench_c2", "bench_c5", "bench_c3_1k", "bench_c3_1k_graph", "bench_dist1", "bench_dist1_bf16"):
    try:
        d = json.load(open("$O/%s.json" % f))
        print(f, "%.2f ms/step  %.0f rays/s  n_gpus %d" % (d["ms_per_step"], d["value"], d["n_gpus"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
rm -rf $O/kt $O/*.log
ls $O
