#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_pfx.py -x -q -m gpu -k "emitter or cfg1 or cfg3_tuples or pfx_tuples or wide_dictionary or product_builder or concurrent or lazy" > $O/r04_c24_pytest.log 2>&1; tail -5 $O/r04_c24_pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o t -- python $R/tools/time_emit.py 1024 sparse 3 > /tmp/pe.log 2>&1
(grep "emit=1" /tmp/pe.log | cut -c1-150; python $R/tools/kstats.py /tmp/pe emit3_expand,emit3_detect,emit3_bin,emit3_desc) > $O/r04_c24_emit.txt 2>&1
rm -rf /tmp/pe2
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe2 -o t -- python $R/tools/time_emit.py 512 dense 3 > /tmp/pe2.log 2>&1
(grep "emit=1" /tmp/pe2.log | cut -c1-150; python $R/tools/kstats.py /tmp/pe2 emit3_expand,emit3_detect,emit3_bin) >> $O/r04_c24_emit.txt 2>&1
cat $O/r04_c24_emit.txt
