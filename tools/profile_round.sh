#!/bin/bash
# Produces the evidence files of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02   ->  gpurun_out/r02_*   (copy what you want judged into profiles/)
# 1. bench.py (full, with cpu_baseline, with_checksum, dense, tuples)   -> <tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of bench.py                        -> <tag>_kernel_stats.csv
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only) -> <tag>_pmc_hbm.json
# 4. SQ counters of the count and the checksum kernel (two passes each)  -> <tag>_pmc_sq.txt
# 5. kernel stats of the tuple emitter                                   -> <tag>_emit_kernel_stats.csv
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python $R/bench.py --op checksum --no-cpu --no-dense --materialize-mib 0 > $OUT/${TAG}_bench_checksum.json 2>> $OUT/${TAG}_bench.err
python $R/bench.py --workload cfg2 --bytes 1073741824 --no-cpu --materialize-mib 0 > $OUT/${TAG}_bench_cfg2.json 2>> $OUT/${TAG}_bench.err
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-dense --materialize-mib 0"
rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -o t -- $BENCH > /tmp/prof_$TAG/trace.log 2>&1
find /tmp/prof_$TAG/trace -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_$TAG/fetch -o f -- $BENCH > /tmp/prof_$TAG/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_$TAG/write -o w -- $BENCH > /tmp/prof_$TAG/write.log 2>&1
python - <<PY
import csv, glob, json, collections
out = {}
for name, pat in (("FETCH_SIZE", "/tmp/prof_$TAG/fetch/**/*counter_collection.csv"), ("WRITE_SIZE", "/tmp/prof_$TAG/write/**/*counter_collection.csv")):
    vals = collections.defaultdict(list)
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                vals[r["Kernel_Name"].split("(")[0][:80]].append(float(r["Counter_Value"]))
    out[name] = {k: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for k, v in vals.items()}
json.dump(out, open("$OUT/${TAG}_pmc_hbm.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -f $OUT/${TAG}_pmc_sq.txt
for v in "0 1" "0 0"; do
  set -- $v
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM"; do
    d=/tmp/prof_$TAG/sq_c$2_$(echo $pass | cut -c4-12)
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $d -o p -- python $R/tools/prof_one.py cfg3 sparse auto 1024 1024 $1 $2 > $d.log 2>&1
    echo "== cfg3 sparse, 1 GiB, engine auto, count_only=$2" >> $OUT/${TAG}_pmc_sq.txt
    python $R/tools/pmc_summary.py $d >> $OUT/${TAG}_pmc_sq.txt 2>&1
  done
done
rm -rf /tmp/prof_$TAG/emit
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/emit -o e -- python $R/tools/time_emit.py 1024 sparse 3 > $OUT/${TAG}_emit.txt 2>&1
find /tmp/prof_$TAG/emit -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_emit_kernel_stats.csv \;
# 6. the restart iterators: cfg3 find_iter / leftmost_find_iter and cfg5 (charwise, true SURVEY 8d workload), timing + kernel stats
for w in sparse dense; do
  rm -rf /tmp/prof_$TAG/find_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/find_$w -o k -- python $R/tools/time_find.py 1024 $w > $OUT/${TAG}_find_$w.txt 2>&1
  find /tmp/prof_$TAG/find_$w -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_find_${w}_kernel_stats.csv \;
done
for m in leftmost find overlapping; do
  python $R/tools/bench_cfg5.py --mode $m > $OUT/${TAG}_bench_cfg5_$m.json 2>> $OUT/${TAG}_bench.err
done
rm -rf /tmp/prof_$TAG/cfg5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/cfg5 -o k -- python $R/tools/bench_cfg5.py --mode leftmost --cpu-mib 0 > /tmp/prof_$TAG/cfg5.log 2>&1
find /tmp/prof_$TAG/cfg5 -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_cfg5_leftmost_kernel_stats.csv \;
cat $OUT/${TAG}_bench.json
head -8 $OUT/${TAG}_kernel_stats.csv
