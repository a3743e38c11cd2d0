#!/bin/bash
# L2 / request counters of the bytewise chain kernels: tools/pmc_find.sh [sparse|dense]
R=${GRAFT_REPO_ROOT:-$(pwd)}; HK=${1:-sparse}
cd /tmp && export TMPDIR=/tmp
export DAAC_PMC_FILTER="chain_kernel<false. 0. 0>"
for pass in "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  d=/tmp/pmcf_$(echo $pass | cut -c1-12 | tr ' ' _)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/time_find.py 1024 $HK > $d.log 2>&1
  python $R/tools/pmc_summary.py $d | grep -v duration_us
done
