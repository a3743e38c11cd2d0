#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p18; mkdir -p /tmp/p18
export DAAC_PMC_FILTER="gram2_kernel"
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=/tmp/p18/sq_$(echo $pass | cut -c1-12)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/prof_one.py cfg3 dense auto 1024 1024 0 1 > $d.log 2>&1
  python $R/tools/pmc_summary.py $d 2>&1 | head -14
  tail -2 $d.log | cut -c1-200
done
