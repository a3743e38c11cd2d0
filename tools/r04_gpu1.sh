#!/bin/bash
# round 4, GPU call 1: the one-detection emitter (emit3_kernels.hip) — parity, A/B against the COUNT + WRITE emitter, kernel breakdown
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tuple_emitter or cfg3_100k or cfg2_1000 or lazy_iterator" > $O/r04_c1_pytest.log 2>&1
tail -15 $O/r04_c1_pytest.log
timeout 300 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg1 or cfg3_tuples or cfg2_full" > $O/r04_c1_pytest2.log 2>&1
tail -5 $O/r04_c1_pytest2.log
for v in 0 1; do timeout 300 python tools/time_emit.py 1024 sparse 3 $v 2>&1 | grep "emit=1"; done | tee $O/r04_c1_emit_ab.txt
timeout 300 python tools/time_emit.py 512 dense 3 0 2>&1 | grep "emit=1" | tee -a $O/r04_c1_emit_ab.txt
timeout 300 python tools/time_emit.py 512 dense 3 1 2>&1 | grep "emit=1" | tee -a $O/r04_c1_emit_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e3 -o t -- python $R/tools/time_emit.py 1024 sparse 3 0 > /tmp/prof_e3.log 2>&1
find /tmp/prof_e3 -name '*kernel_stats.csv' -exec cp {} $O/r04_c1_emit_kernel_stats.csv \;
head -12 $O/r04_c1_emit_kernel_stats.csv
