#!/bin/bash
# Evidence of a round on the GPU box (through gpurun from the repo root):  tools/profile_round.sh r03  ->  gpurun_out/r03_*
# (copy what is to be judged into profiles/).  Every workload that has a bench line gets: the line, a kernel-time summary over the
# TIMED launches only (tools/trace_summary.py drops the warm-ups), and HBM traffic from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)
# gathered in <tag>_hbm_traffic.json, which bench.py and the side benches read for `roofline.traffic`.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
P=/tmp/prof_$TAG
mkdir -p $OUT
rm -rf $P && mkdir -p $P
cd /tmp && export TMPDIR=/tmp
echo '{}' > $OUT/${TAG}_hbm_traffic.json

# one workload: name, kernel-name filter, warm-up launches to drop, command...
profile() {
  local name=$1 filter=$2 drop=$3; shift 3
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/${name}_trace -o t -- "$@" > $P/${name}_trace.log 2>&1
  find $P/${name}_trace -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_${name}_kernel_stats_all_launches.csv \;
  python $R/tools/trace_summary.py $P/${name}_trace $drop > $OUT/${TAG}_${name}_kernel_stats.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/${name}_fetch -o f -- "$@" > $P/${name}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/${name}_write -o w -- "$@" > $P/${name}_write.log 2>&1
  python $R/tools/pmc_hbm.py $P/${name}_fetch $P/${name}_write $drop "$filter" > $P/${name}_hbm.json
  python - <<PY
import json
t = json.load(open("$OUT/${TAG}_hbm_traffic.json"))
t["$name"] = json.load(open("$P/${name}_hbm.json"))
json.dump(t, open("$OUT/${TAG}_hbm_traffic.json", "w"), indent=1)
PY
  echo "== $name"; head -4 $OUT/${TAG}_${name}_kernel_stats.csv; cat $P/${name}_hbm.json | head -12
}

B="python $R/bench.py --steps 8 --warmup 2 --no-cpu --no-dense --materialize-mib 0"
profile cfg3_sparse_count   "gram3_kernel,gram2_kernel,gram_count_kernel" 2 $B
profile cfg3_sparse_checksum "gram_count_kernel,gram2_kernel" 2 $B --op checksum
profile cfg3_dense_count    "gram3_kernel" 2 $B --haystack dense
profile cfg2_count          "gram3_kernel" 2 $B --workload cfg2 --bytes 1073741824
profile emit                "gram2_emit_kernel,exclusive" 2 python $R/tools/time_emit.py 1024 sparse 3
profile find_sparse         "chain,restart" 0 python $R/tools/time_find.py 1024 sparse
profile find_dense          "chain,restart" 0 python $R/tools/time_find.py 1024 dense
profile cfg5_leftmost       "char" 0 python $R/tools/bench_cfg5.py --mode leftmost --cpu-mib 0
profile cfg5_find           "char" 0 python $R/tools/bench_cfg5.py --mode find --cpu-mib 0
profile cfg5_overlapping    "char" 0 python $R/tools/bench_cfg5.py --mode overlapping --cpu-mib 0
profile any_alphabet        "pfx_kernel" 2 python $R/tools/ab_pfx.py 1024

# SQ counters of the count and the checksum kernel (two passes each)
rm -f $OUT/${TAG}_pmc_sq.txt
for v in "0 1" "0 0"; do
  set -- $v
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"; do
    d=$P/sq_c$2_$(echo $pass | cut -c4-12)
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/prof_one.py cfg3 sparse auto 1024 1024 $1 $2 > $d.log 2>&1
    echo "== cfg3 sparse, 1 GiB, engine auto, count_only=$2" >> $OUT/${TAG}_pmc_sq.txt
    python $R/tools/pmc_summary.py $d >> $OUT/${TAG}_pmc_sq.txt 2>&1
  done
done

# the bench lines themselves (they read <tag>_hbm_traffic.json for roofline.traffic when it sits in profiles/; here: from gpurun_out)
export DAAC_HBM_TRAFFIC_JSON=$OUT/${TAG}_hbm_traffic.json
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python $R/bench.py --op checksum --no-cpu --no-dense --materialize-mib 0 > $OUT/${TAG}_bench_cfg3_checksum.json 2>> $OUT/${TAG}_bench.err
python $R/bench.py --haystack dense --no-cpu --no-dense --materialize-mib 0 > $OUT/${TAG}_bench_cfg3_dense.json 2>> $OUT/${TAG}_bench.err
python $R/bench.py --workload cfg2 --bytes 1073741824 --no-cpu --materialize-mib 0 > $OUT/${TAG}_bench_cfg2.json 2>> $OUT/${TAG}_bench.err
for m in leftmost find overlapping; do
  python $R/tools/bench_cfg5.py --mode $m > $OUT/${TAG}_bench_cfg5_$m.json 2>> $OUT/${TAG}_bench.err
done
python $R/tools/time_find.py 1024 sparse > $OUT/${TAG}_find_sparse.txt 2>&1
python $R/tools/time_find.py 1024 dense > $OUT/${TAG}_find_dense.txt 2>&1
python $R/tools/time_emit.py 1024 sparse 3 > $OUT/${TAG}_emit.txt 2>&1
python $R/tools/ab_pfx.py 1024 > $OUT/${TAG}_any_alphabet.txt 2>&1
cat $OUT/${TAG}_bench.json | head -c 1500
