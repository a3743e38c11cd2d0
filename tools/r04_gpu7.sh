#!/bin/bash
# emit3 EXPAND: start delay between the waves of a CU (emit_stagger) — timing sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for st in 0 1 2 3 4 6 8; do
  rm -rf /tmp/pe_s
  DAAC_OPT_emit_stagger=$st timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_s -o t -- python $R/tools/time_emit.py 1024 sparse 3 0 only16 > /tmp/pe_s.log 2>&1
  echo "== stagger $st"; grep "emit=1" /tmp/pe_s.log | cut -c1-140
  python $R/tools/kstats.py /tmp/pe_s emit3_expand,emit3_detect
done > $O/r04_c7_stagger.txt 2>&1
cat $O/r04_c7_stagger.txt
