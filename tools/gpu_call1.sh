#!/bin/bash
# Round-2 GPU call 1: full -m gpu test run, bench line, kernel trace, PMC passes (traffic, MFMA), FETCH_SIZE calibration.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2c1
mkdir -p $O
python -m pytest tests -m gpu -q -s -rfE --tb=short > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -n 25 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; head -c 1500 $O/bench.json
B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o kt -- $B > $GRAFT_REPO_ROOT/$O/kt.log 2>&1 )
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 60 > $O/kernel_stats.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE|SQ_ACTIVE_INST_VALU|SQ_INSTS_VALU " | head -60 > $O/counters.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAVE_CYCLES"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$T -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant-steps 0 --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_$T.log 2>&1 )
  python tools/rocpd_pmc.py $(find $O/pmc_$T -name "*.db" | head -1) > $O/pmc_$T.txt 2>&1
done
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/$O/calib -o c -- python $GRAFT_REPO_ROOT/tools/gather_calib.py > $GRAFT_REPO_ROOT/$O/calib.log 2>&1 )
python tools/rocpd_pmc.py $(find $O/calib -name "*.db" | head -1) > $O/calib_pmc.txt 2>&1
rm -rf $O/kt $O/pmc_*/ $O/calib   # databases are large; the text summaries stay
ls -la $O
