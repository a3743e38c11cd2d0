#!/bin/bash
# round 4, GPU call 3: emit3 with the LDS ring + dense copy-out and aggregated atomics — parity first, then timing of shipped + variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
L=$R/daachorse_amd/lib/libdaachorse_amd.so
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tuple_emitter or cfg3_100k or cfg2_1000 or lazy_iterator" > $O/r04_c18_pytest.log 2>&1
tail -4 $O/r04_c18_pytest.log
timeout 300 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg1 or cfg3_tuples or cfg2_full" > $O/r04_c18_pytest2.log 2>&1
tail -3 $O/r04_c18_pytest2.log
cd /tmp && export TMPDIR=/tmp
cp $L /tmp/_orig.so
one() {  # name mib kind
  rm -rf /tmp/pe_$1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$1 -o t -- python $R/tools/time_emit.py $2 $3 3 0 > /tmp/pe_$1.log 2>&1
  echo "== $1 $2 $3"; grep "emit=1" /tmp/pe_$1.log
  python $R/tools/kstats.py /tmp/pe_$1 emit3,exclusive,scan_chunk
}
one shipped 1024 sparse > $O/r04_c18_variants.txt 2>&1
one shipped_dense 512 dense >> $O/r04_c18_variants.txt 2>&1
for f in $R/abtmp/lib_e3_*.so; do cp $f $L; one $(basename $f .so) 1024 sparse >> $O/r04_c18_variants.txt 2>&1; done
cp /tmp/_orig.so $L
cat $O/r04_c18_variants.txt
