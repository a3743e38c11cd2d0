#!/bin/bash
# restart-iterator kernel stats (step 6 of profile_round.sh alone) + facade test
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=r02b; mkdir -p $OUT
timeout 300 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "facade" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for w in sparse; do
  rm -rf /tmp/prof_$TAG/find_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/find_$w -o k -- python $R/tools/time_find.py 1024 $w > $OUT/${TAG}_find_$w.txt 2>&1
  find /tmp/prof_$TAG/find_$w -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_find_${w}_kernel_stats.csv \;
done
rm -rf /tmp/prof_$TAG/cfg5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/cfg5 -o k -- python $R/tools/bench_cfg5.py --mode leftmost --cpu-mib 0 > $OUT/${TAG}_cfg5.log 2>&1
find /tmp/prof_$TAG/cfg5 -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_cfg5_leftmost_kernel_stats.csv \;
tail -3 $OUT/${TAG}_cfg5.log; cut -c1-150 $OUT/${TAG}_cfg5_leftmost_kernel_stats.csv | head -12
grep -v "^W\|^E" $OUT/${TAG}_find_sparse.txt | tail -5; cut -c1-150 $OUT/${TAG}_find_sparse_kernel_stats.csv | head -14
