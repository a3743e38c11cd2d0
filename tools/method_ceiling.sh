#!/bin/bash
# The measured ceiling of the GRAM method's main path (roofline.method_ceiling of bench.py): the `.count()` kernel with everything behind the
# per-position lookup compiled out (hits never queued: -DG4X=1 of gram4_kernels.hip; its count is WRONG on purpose), timed by bench.py
# itself on the headline workload.  Through gpurun from the repo root:  tools/method_ceiling.sh r05
#   needs abtmp/lib_g4x1.so  (tools/mkvar5.sh g4x1 gram4_kernels -DG4X=1)
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/daachorse_amd/lib/libdaachorse_amd.so
cp $L /tmp/_orig.so
cp $R/abtmp/lib_g4x1.so $L
python $R/bench.py --steps 8 --warmup 2 --no-cpu --no-dense --no-extra --materialize-mib 0 > /tmp/ceil_sparse.json 2>/dev/null
python $R/bench.py --steps 8 --warmup 2 --no-cpu --no-dense --no-extra --materialize-mib 0 --haystack dense > /tmp/ceil_dense.json 2>/dev/null
cp /tmp/_orig.so $L
python - <<PY
import json
s = json.load(open("/tmp/ceil_sparse.json")); d = json.load(open("/tmp/ceil_dense.json"))
out = {"what": "gram4 .count() kernel with the hit path compiled out (main path only: byte -> class -> K-gram index -> one M word per position, "
               "counts of the short patterns, hit bits computed and dropped); 4 GiB, HIP events over 8 launches; the count it returns is wrong on purpose",
       "cfg3_sparse": {"GB/s": s["roofline"]["achieved"], "frac": s["roofline"]["frac"], "kernel_ms": s["roofline"]["kernel_ms"]},
       "cfg3_dense": {"GB/s": d["roofline"]["achieved"], "frac": d["roofline"]["frac"], "kernel_ms": d["roofline"]["kernel_ms"]},
       "build": "tools/mkvar5.sh g4x1 gram4_kernels -DG4X=1", "round": "$TAG"}
json.dump(out, open("$R/gpurun_out/${TAG}_method_ceiling.json", "w"), indent=1)
print(json.dumps(out))
PY
