#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for il in 1 0; do
  rm -rf /tmp/fx_$il
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fx_$il -o f -- python $R/tools/bench_cfg5.py --mode leftmost --cpu-mib 0 --steps 2 --opt interleave=$il > /tmp/fx_$il.log 2>&1
  echo "== interleave=$il"; grep -o '"value": [0-9.]*' /tmp/fx_$il.log | head -1
  python $R/tools/pmc_hbm.py /tmp/fx_$il /tmp/fx_$il 0 "char_chain,interleave" | grep -E "FETCH|\"void|\"daac" | head -12
  python $R/tools/trace_summary.py /tmp/fx_$il 0 "char_chain,interleave" | cut -c1-120
done
