cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pe; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o t -- python $R/tools/time_emit.py 1024 sparse 2 > /tmp/pe.log 2>&1
grep -v amdgpu.ids /tmp/pe.log | tail -3
f=$(find /tmp/pe -name '*kernel_stats.csv' | head -1); head -12 $f | cut -c1-200
