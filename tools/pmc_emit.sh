#!/bin/bash
# the emitter's SQ counters alone (the part of tools/profile_round.sh whose kernel filter was wrong in the first run)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; P=/tmp/prof_sq; rm -rf $P; mkdir -p $P; cd /tmp; export TMPDIR=/tmp
PASS1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
rm -f $OUT/r04_pmc_sq_emit.txt
for pass in "$PASS1" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$P/sq_emit_$(echo $pass | cut -c4-12)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/time_emit.py 1024 sparse 3 0 only16 > $d.log 2>&1
  echo "== tuple emitter, cfg3 sparse, 1 GiB, 16-byte tuples" >> $OUT/r04_pmc_sq_emit.txt
  DAAC_PMC_FILTER=emit3_detect,emit3_bin,emit3_expand python $R/tools/pmc_summary.py $d | grep -v duration_us >> $OUT/r04_pmc_sq_emit.txt 2>&1
done
cat $OUT/r04_pmc_sq_emit.txt
