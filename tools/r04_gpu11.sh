#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $O/r04_c11_pytest_full.log 2>&1
tail -12 $O/r04_c11_pytest_full.log
timeout 1500 python bench.py > $O/r04_c11_bench.json 2> $O/r04_c11_bench.err
tail -5 $O/r04_c11_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_c11_bench.json"))
for k in ("value", "roofline", "with_checksum", "dense", "tuples_device", "tuples_device_24", "materialize", "restart", "cfg5", "iterator"):
    print(k, json.dumps(d.get(k))[:900])
PY
