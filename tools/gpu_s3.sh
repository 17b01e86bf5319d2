#!/bin/bash
# round-5 session 3: static-grid pre-pass next to the LDS dynamic-hash kernel (side stream) + counters of the split encode
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 500 bash tools/gpu_ab.sh s3 none default side:L4D_STREAMS=4 nosplit:L4D_ENC_HS_SPLIT=0
timeout 300 bash tools/gpu_pmc_quick.sh s3 l2:TCC_REQ_sum,TCC_HIT_sum,TCC_MISS_sum sq:SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU,SQ_INSTS_VMEM_RD
grep -E "density_encode|hashgrid_fwd_levels|hashgrid_t_fwd" gpurun_out/s3/pmc_l2.txt gpurun_out/s3/pmc_sq.txt | cut -c1-250
