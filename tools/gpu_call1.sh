#!/bin/bash
# round 3, GPU call 1: gram3 correctness + A/B sweep + SQ counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_gram3.py -x -q > $OUT/c1_gram3_tests.txt 2>&1
tail -5 $OUT/c1_gram3_tests.txt
timeout 600 python tools/ab_count.py 1024 > $OUT/c1_ab_count.txt 2>&1
cat $OUT/c1_ab_count.txt
cd /tmp && export TMPDIR=/tmp
mkdir -p /tmp/prof_c1
rm -f $OUT/c1_pmc_sq.txt
for v in 2 3; do
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"; do
    d=/tmp/prof_c1/sq_v${v}_$(echo $pass | cut -c4-12)
    rm -rf $d
    timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/prof_one.py cfg3 sparse gram 1024 1024 $v 1 > $d.log 2>&1
    echo "== cfg3 sparse, 1 GiB, gram_version=$v, count only" >> $OUT/c1_pmc_sq.txt
    python $R/tools/pmc_summary.py $d >> $OUT/c1_pmc_sq.txt 2>&1
  done
done
cat $OUT/c1_pmc_sq.txt
