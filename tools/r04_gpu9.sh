#!/bin/bash
# SQ counters of the emit3 kernels (1 GiB cfg3, 16-byte tuples)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -f $O/r04_c9_pmc.txt
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  d=/tmp/pmc_$(echo $pass | cut -c4-16 | tr ' ' '_')
  rm -rf $d
  timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/time_emit.py 1024 sparse 1 0 only16 > $d.log 2>&1
  echo "== $pass" >> $O/r04_c9_pmc.txt
  DAAC_PMC_FILTER=emit3_expand,emit3_detect python $R/tools/pmc_summary.py $d 2>&1 | grep -v duration_us | cut -c1-160 >> $O/r04_c9_pmc.txt
done
cat $O/r04_c9_pmc.txt
