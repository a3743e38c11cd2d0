#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 3 --warmup 1 --materialize-mib 0 --no-extra --cpu-seconds 2 > gpurun_out/r04_c19_bench.json 2> gpurun_out/r04_c19_bench.err; tail -3 gpurun_out/r04_c19_bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04_c19_bench.json').read().strip().splitlines()[-1])
for k,v in b.get('any_alphabet',{}).items():
    print(k, json.dumps({x:v[x] for x in v if x!='dictionary'}))
PY
