#!/bin/bash
# round 4, GPU call 2: where the time of emit3 goes — timing-only variants (wrong tuples on purpose) and SQ / HBM counters of the shipped kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
L=$R/daachorse_amd/lib/libdaachorse_amd.so
cd /tmp && export TMPDIR=/tmp
cp $L /tmp/_orig.so
one() {  # name
  rm -rf /tmp/pe_$1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$1 -o t -- python $R/tools/time_emit.py 1024 sparse 3 0 only16 > /tmp/pe_$1.log 2>&1
  echo "== $1"; grep "emit=1" /tmp/pe_$1.log
  find /tmp/pe_$1 -name '*kernel_stats.csv' -exec cat {} \; | grep -E "emit3|exclusive|scan_chunk" | awk -F, '{print $1, $2, $4}' | cut -c1-150
}
one shipped > $O/r04_c2_variants.txt 2>&1
for f in $R/abtmp/lib_e3_*.so; do cp $f $L; one $(basename $f .so) >> $O/r04_c2_variants.txt 2>&1; done
cp /tmp/_orig.so $L
cat $O/r04_c2_variants.txt
# counters of the shipped kernels
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  d=/tmp/pmc_$(echo $pass | cut -c1-14 | tr ' ' '_')
  rm -rf $d
  timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/time_emit.py 1024 sparse 1 0 only16 > $d.log 2>&1
  echo "== $pass" >> $O/r04_c2_pmc.txt
  DAAC_PMC_FILTER=emit3 python $R/tools/pmc_summary.py $d 2>&1 | cut -c1-200 >> $O/r04_c2_pmc.txt
done
cat $O/r04_c2_pmc.txt | head -80
