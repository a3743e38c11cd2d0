#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -m pytest tests/test_gpu_charwise.py tests/test_gpu_configs.py -x -q -k "not 4_gib and not window_counts" 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py -x -q -k "find or leftmost or chain or beyond" 2>&1 | tail -4
for il in 1 0; do
  echo "== interleave=$il"
  for m in leftmost find overlapping; do python tools/bench_cfg5.py --mode $m --cpu-mib 0 --opt interleave=$il 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['metric'][:60], d['value'], 'GB/s', d['ms_per_step'], 'ms', d['match_count'])"; done
done
python tools/time_find.py 1024 sparse 2>&1 | grep -v amdgpu
