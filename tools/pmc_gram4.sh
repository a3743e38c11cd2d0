#!/bin/bash
# SQ instruction counters of the `.count()` kernel (gram4) for the shipped library and the stage builds abtmp/lib_g4x*.so (stages compiled
# out, counts wrong on purpose): cfg3, 1 GiB.   bash tools/pmc_gram4.sh [sparse|dense] [extra env: DAAC_OPTS="gram_ppl=32,gram2_rfull=0"]
R=${GRAFT_REPO_ROOT:-$(pwd)}; HK=${1:-sparse}
L=$R/daachorse_amd/lib/libdaachorse_amd.so
cd /tmp && export TMPDIR=/tmp
export DAAC_PMC_FILTER="gram4_kernel"
cp $L /tmp/_orig.so
for f in shipped $R/abtmp/lib_g4x*.so; do
  [ $f != shipped ] && cp $f $L
  echo "=== $(basename $f)"
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
    d=/tmp/pmcd; rm -rf $d
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o p -- python $R/tools/prof_one.py cfg3 $HK auto 1024 1024 0 1 > $d.log 2>&1
    python $R/tools/pmc_summary.py $d | grep -v "duration_us\|^/tmp"
    python $R/tools/pmc_summary.py $d | grep "duration_us" | awk '{s+=$2; n++} END {if (n) printf "   mean duration_us %.1f (n=%d)\n", s/n, n}'
  done
done
cp /tmp/_orig.so $L
