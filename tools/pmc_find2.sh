#!/bin/bash
# SQ counters of the bytewise chain kernels for the shipped library and every abtmp/lib_*.so
R=${GRAFT_REPO_ROOT:-$(pwd)}; HK=${1:-sparse}
L=$R/daachorse_amd/lib/libdaachorse_amd.so
cd /tmp && export TMPDIR=/tmp
export DAAC_PMC_FILTER="chain_kernel<false. 0. 0>,chain_kernel<true. 0. 0>"
cp $L /tmp/_orig.so
for f in shipped $R/abtmp/lib_*.so; do
  [ $f != shipped ] && cp $f $L
  echo "=== $f"
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_LEVEL_WAVES"; do
    d=/tmp/pmcg
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/time_find.py 1024 $HK > $d.log 2>&1
    python $R/tools/pmc_summary.py $d | grep -v "lds 0\|^   duration_us.*vgpr" 
    python $R/tools/pmc_summary.py $d | grep "duration_us" | sort | uniq -c | head -4
  done
done
cp /tmp/_orig.so $L
