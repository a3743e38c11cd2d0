#!/bin/bash
# Produces the evidence files of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01   ->  gpurun_out/r01_*   (copy what you want judged into profiles/)
# 1. bench.py (full, with cpu_baseline)            -> <tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of bench.py  -> <tag>_kernel_stats.csv
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no trace domains besides kernels)
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python $R/bench.py --haystack dense --no-cpu > $OUT/${TAG}_bench_dense.json 2>> $OUT/${TAG}_bench.err
python $R/bench.py --workload cfg2 --bytes 1073741824 --no-cpu > $OUT/${TAG}_bench_cfg2.json 2>> $OUT/${TAG}_bench.err
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu --materialize-mib 0"
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
cat $OUT/${TAG}_bench.json
head -8 $OUT/${TAG}_kernel_stats.csv
