#!/bin/bash
# Evidence of a round on the GPU box (through gpurun from the repo root):  tools/profile_round.sh r04  ->  gpurun_out/r04_*
# (copy what is to be judged into profiles/).  Every workload that has a bench object gets: a kernel-time summary over the TIMED launches
# only (tools/trace_summary.py drops the warm-ups) and HBM traffic from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) gathered in
# <tag>_hbm_traffic.json — stamped with the hash of the sources the library was built from — which bench.py reads for `roofline.traffic`,
# `traffic_stale` and the tuple emitter's write-traffic ratio.  Then SQ counters of the count / checksum / emitter kernels, and the bench
# line itself (which carries the restart iterators, cfg5, the lazy iterator, the wide dictionaries) plus its variants.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
P=/tmp/prof_$TAG
mkdir -p $OUT
rm -rf $P && mkdir -p $P
cd /tmp && export TMPDIR=/tmp
HASH=$(cd $R && python -c "from daachorse_amd import _build; print(_build.source_hash())")
SO=$(sha256sum $R/daachorse_amd/lib/libdaachorse_amd.so | cut -d' ' -f1)
python - <<PY
import json
json.dump({"_csrc_sha256": "$HASH", "_so_sha256": "$SO",
           "_how": "tools/profile_round.sh $TAG: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over each workload's own command (the bench line's command for cfg3/cfg2), warm-up launches dropped (tools/pmc_hbm.py); KiB per launch; bytes_per_launch = 2*FETCH + WRITE: FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction (re-calibrated here: profiles/r01_fetch_size_calibration.txt), WRITE_SIZE 1:1. The doubling over-counts whatever part of the fetches is not a coalesced stream (record gathers, walker slabs).",
           "_source": "profiles/${TAG}_hbm_traffic.json, one MI355X; a figure from a separate profiling run of the same command, not measured inside the run that prints it"},
          open("$OUT/${TAG}_hbm_traffic.json", "w"), indent=1)
PY

# one workload: name, kernel-name filter, warm-up launches to drop, command...
profile() {
  local name=$1 filter=$2 drop=$3; shift 3
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/${name}_trace -o t -- "$@" > $P/${name}_trace.log 2>&1
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
  echo "== $name"; head -5 $OUT/${TAG}_${name}_kernel_stats.csv; cat $P/${name}_hbm.json | head -14
}

B="python $R/bench.py --steps 8 --warmup 2 --no-cpu --no-dense --materialize-mib 0 --no-extra"
profile cfg3_sparse_count    "gram4_kernel,gram2_kernel,gram_count_kernel" 2 $B
profile cfg3_sparse_checksum "gram_count_kernel,gram2_kernel" 2 $B --op checksum
profile cfg3_dense_count     "gram4_kernel" 2 $B --haystack dense
profile cfg2_count           "gram4_kernel" 2 $B --workload cfg2 --bytes 1073741824
profile emit                 "emit3_detect_kernel,emit3_bin_kernel,emit3_expand_kernel" 2 python $R/tools/time_emit.py 1024 sparse 3
profile emit_dense           "emit3_detect_kernel,emit3_bin_kernel,emit3_expand_kernel" 2 python $R/tools/time_emit.py 512 dense 3
profile find_sparse          "find3_,emit3_detect,emit3_bin,chain,restart" 0 python $R/tools/time_find.py 1024 sparse find
profile find_dense           "chain,restart" 0 python $R/tools/time_find.py 1024 dense find
profile leftmost_sparse      "left3_,emit3_detect,emit3_bin,chain,restart" 0 python $R/tools/time_find.py 1024 sparse leftmost
profile leftmost_dense       "chain,restart" 0 python $R/tools/time_find.py 1024 dense leftmost
profile cfg5_leftmost        "char" 0 python $R/tools/bench_cfg5.py --mode leftmost --cpu-mib 0
profile cfg5_find            "char" 0 python $R/tools/bench_cfg5.py --mode find --cpu-mib 0
profile cfg5_overlapping     "char" 0 python $R/tools/bench_cfg5.py --mode overlapping --cpu-mib 0
profile any_alphabet         "pfx_kernel" 2 python $R/tools/ab_pfx.py 1024

# SQ counters: the count and the checksum kernel (two passes each), then the emitter's three kernels
rm -f $OUT/${TAG}_pmc_sq.txt
PASS1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
PASS2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"
for v in "0 1" "0 0"; do
  set -- $v
  for pass in "$PASS1" "$PASS2"; do
    d=$P/sq_c$2_$(echo $pass | cut -c4-12)
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/prof_one.py cfg3 sparse auto 1024 1024 $1 $2 > $d.log 2>&1
    echo "== cfg3 sparse, 1 GiB, engine auto, count_only=$2" >> $OUT/${TAG}_pmc_sq.txt
    python $R/tools/pmc_summary.py $d >> $OUT/${TAG}_pmc_sq.txt 2>&1
  done
done
for pass in "$PASS1" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$P/sq_emit_$(echo $pass | cut -c4-12)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/time_emit.py 1024 sparse 3 0 only16 > $d.log 2>&1
  echo "== tuple emitter, cfg3 sparse, 1 GiB, 16-byte tuples" >> $OUT/${TAG}_pmc_sq.txt
  DAAC_PMC_FILTER=emit3_detect,emit3_bin,emit3_expand python $R/tools/pmc_summary.py $d | grep -v duration_us >> $OUT/${TAG}_pmc_sq.txt 2>&1
done

# the selection kernels of the restart iterators (find3 / left3) with the emitter's front half they run behind
for what in find leftmost; do
  for pass in "$PASS1" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    d=$P/sq_${what}_$(echo $pass | cut -c4-12)
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/time_find.py 1024 sparse $what > $d.log 2>&1
    echo "== ${what}_iter by selection, cfg3 sparse, 1 GiB (count + checksum)" >> $OUT/${TAG}_pmc_sq.txt
    DAAC_PMC_FILTER=find3_,left3_,emit3_detect,emit3_bin python $R/tools/pmc_summary.py $d | grep -v duration_us >> $OUT/${TAG}_pmc_sq.txt 2>&1
  done
done

# BASELINE configs[4] (charwise leftmost_find_iter) on the chain walkers, and what a position-parallel route over the same text would start
# from: the PFX count / count + checksum kernels on cfg5's patterns scanned bytewise (utf8jp)
for pass in "$PASS1" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"; do
  d=$P/sq_cfg5_$(echo $pass | cut -c4-12)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/bench_cfg5.py --mode leftmost --cpu-mib 0 > $d.log 2>&1
  echo "== cfg5: charwise leftmost_find_iter (LeftmostLongest), 1 GiB, chain walkers" >> $OUT/${TAG}_pmc_sq.txt
  DAAC_PMC_FILTER=char_chain python $R/tools/pmc_summary.py $d | grep -v duration_us >> $OUT/${TAG}_pmc_sq.txt 2>&1
  d=$P/sq_utf8jp_$(echo $pass | cut -c4-12)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/ab_pfx.py 1024 > $d.log 2>&1
  echo "== cfg5's patterns scanned bytewise (utf8jp) on PFX: .count() and count + checksum, 1 GiB of the same text" >> $OUT/${TAG}_pmc_sq.txt
  DAAC_PMC_FILTER="pfx_kernel<6" python $R/tools/pmc_summary.py $d | grep -v duration_us >> $OUT/${TAG}_pmc_sq.txt 2>&1
done

# the `.count()` kernel stage by stage (needs abtmp/lib_g4x1..5.so: tools/mkvar5.sh g4xN gram4_kernels -DG4X=N) and the main path alone under bench.py
if ls $R/abtmp/lib_g4x1.so > /dev/null 2>&1; then
  (cd $R && bash tools/ab_libs3.sh 4096 tools/ab_count_shapes.json 2>&1 | grep "cfg\|==") > $OUT/${TAG}_gram4_decomposition.txt
  (cd $R && bash tools/pmc_gram4.sh sparse) > $OUT/${TAG}_pmc_decomp_sparse.txt 2>&1
  (cd $R && bash tools/pmc_gram4.sh dense) > $OUT/${TAG}_pmc_decomp_dense.txt 2>&1
  (cd $R && bash tools/method_ceiling.sh $TAG) > $P/method_ceiling.log 2>&1
  cd /tmp
fi

# the wide look-alike dictionaries' kernels (SQ + L2 counters) and what a turn of the charwise walkers does on cfg5 (needs abtmp/cwprof: tools/cw_taxonomy.sh build)
(cd $R && bash tools/pmc_wide.sh 256) > $OUT/${TAG}_pmc_wide.txt 2>&1
if [ -f $R/abtmp/cwprof/libdaachorse_amd.so ]; then (cd $R && bash tools/cw_taxonomy.sh run 1024) > $OUT/${TAG}_cfg5_turn_taxonomy.txt 2>&1; fi
cd /tmp

# the bench lines themselves (they read the traffic file given here; in the repository: profiles/hbm_traffic.json)
export DAAC_HBM_TRAFFIC_JSON=$OUT/${TAG}_hbm_traffic.json
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python $R/bench.py --op checksum --no-cpu --no-dense --materialize-mib 0 --no-extra > $OUT/${TAG}_bench_cfg3_checksum.json 2>> $OUT/${TAG}_bench.err
python $R/bench.py --haystack dense --no-cpu --no-dense --materialize-mib 0 --no-extra > $OUT/${TAG}_bench_cfg3_dense.json 2>> $OUT/${TAG}_bench.err
python $R/bench.py --workload cfg2 --bytes 1073741824 --no-cpu --materialize-mib 0 --no-extra > $OUT/${TAG}_bench_cfg2.json 2>> $OUT/${TAG}_bench.err
python $R/tools/time_find.py 1024 sparse > $OUT/${TAG}_find_sparse.txt 2>&1
python $R/tools/time_find.py 1024 dense > $OUT/${TAG}_find_dense.txt 2>&1
python $R/tools/time_find_tuples.py 1024 sparse > $OUT/${TAG}_find_tuples.txt 2>&1
DAAC_OPT_UNUSED=1 python - > $OUT/${TAG}_find_tuples_walkers.txt 2>&1 <<PY
import sys, runpy
sys.path.insert(0, "$R")
import daachorse_amd as da
da.set_option("select_emit", 0)
sys.argv = ["time_find_tuples.py", "1024", "sparse"]
runpy.run_path("$R/tools/time_find_tuples.py", run_name="__main__")
PY
python $R/tools/time_emit.py 1024 sparse 3 > $OUT/${TAG}_emit.txt 2>&1
python $R/tools/time_emit.py 512 dense 3 >> $OUT/${TAG}_emit.txt 2>&1
python $R/tools/time_iter.py 1024 > $OUT/${TAG}_iterator.txt 2>&1
python $R/tools/ab_wide.py 1024 > $OUT/${TAG}_wide_dictionaries.txt 2>&1
cat $OUT/${TAG}_bench.json | head -c 1500
