mkdir -p gpurun_out/c3
O=$PWD/gpurun_out/c3
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -x -q -k "gram2 or golden_vectors_overlapping or fuzz_small or cfg2 or cfg3 or shard or count_engines" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log)
tail -4 $O/pytest.log
timeout 600 python tools/sweep.py --workload cfg3 --haystack sparse --mib 2048 --reps 8 --grid "engine=gram;gram_version=1,2;count_only=0,1" 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
timeout 600 python tools/sweep.py --workload cfg3 --haystack sparse --mib 2048 --reps 8 --grid "engine=gram;gram_version=2;count_only=1;gram2_rfull=0,1;gram_region=8192,16384,65536" 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
timeout 600 python tools/sweep.py --workload cfg3 --haystack dense --mib 2048 --reps 8 --grid "engine=gram;gram_version=1,2;count_only=0,1" 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "2 1"; do
  set -- $v
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"; do
    d=/tmp/pmc_v$1_c$2_$(echo $pass | cut -c4-12)
    rm -rf $d
    timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/prof_one.py cfg3 sparse gram 1024 1024 $1 $2 > $d.log 2>&1
    echo "== gram_version=$1 count_only=$2" >> $O/sq.txt
    python $R/tools/pmc_summary.py $d >> $O/sq.txt 2>&1
  done
done
cat $O/sq.txt | head -60
