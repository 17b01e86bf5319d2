cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04e; mkdir -p $O
( cd /tmp && timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -d $GRAFT_REPO_ROOT/$O/cal -o c -- $GRAFT_REPO_ROOT/tools/ubench/calib_ubench > $GRAFT_REPO_ROOT/$O/calib.log 2>&1 )
python tools/rocpd_pmc.py $(find $O/cal -name "*.db" | head -1) > $O/calib_by_size.txt 2>&1
rm -rf $O/cal; cat $O/calib_by_size.txt | cut -c1-150
