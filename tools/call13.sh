#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p13; mkdir -p /tmp/p13
export DAAC_PMC_FILTER="chain_kernel<true, 0"
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM"; do
  d=/tmp/p13/sq_$(echo $pass | cut -c4-12)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/bench_cfg5.py --mode leftmost --cpu-mib 0 --steps 2 > $d.log 2>&1
  python $R/tools/pmc_summary.py $d 2>&1 | head -14
done
