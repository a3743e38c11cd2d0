#!/bin/bash
# SQ instruction counters of the `.count()` kernel for the shipped library and the decomposition builds abtmp/lib_g3_*.so (stages compiled out,
# counts wrong on purpose): cfg3 uniform text, 1 GiB.   bash tools/pmc_decomp.sh [sparse|dense]
R=${GRAFT_REPO_ROOT:-$(pwd)}; HK=${1:-sparse}
L=$R/daachorse_amd/lib/libdaachorse_amd.so
cd /tmp && export TMPDIR=/tmp
export DAAC_PMC_FILTER="gram3_kernel"
cp $L /tmp/_orig.so
for f in shipped $R/abtmp/lib_g3_*.so; do
  [ $f != shipped ] && cp $f $L
  echo "=== $(basename $f)"
  d=/tmp/pmcd; rm -rf $d
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $d -o p -- python $R/tools/prof_one.py cfg3 $HK auto 1024 1024 0 1 > $d.log 2>&1
  python $R/tools/pmc_summary.py $d | grep -v "^   duration_us.*vgpr"
  python $R/tools/pmc_summary.py $d | grep "duration_us" | awk '{s+=$2; n++} END {if (n) printf "   mean duration_us %.1f (n=%d)\n", s/n, n}'
done
cp /tmp/_orig.so $L
