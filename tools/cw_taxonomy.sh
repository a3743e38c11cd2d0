#!/bin/bash
# What a turn of the charwise chain walkers does on cfg5 (per symbol: turns, probes, probe hits, failure links followed, symbols settled by
# ROOT's row, dead walks, reports; bytes per symbol walked): the library with tools/variants/cw_turn_taxonomy.patch applied to copies of
# chain_scan.hpp / charwise_kernels.hip (counters added with atomics: timing means nothing in this build).
#   here:            bash tools/cw_taxonomy.sh build      -> abtmp/cwprof/libdaachorse_amd.so   (needs daachorse_amd/build/)
#   through gpurun:  bash tools/cw_taxonomy.sh run [mib]
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$R/abtmp/cwprof
if [ "$1" = build ]; then
  set -e
  mkdir -p $V/s && cp $R/daachorse_amd/csrc/chain_scan.hpp $R/daachorse_amd/csrc/charwise_kernels.hip $V/s/
  sed "s#daachorse_amd/csrc/##; s#abtmp/cwprof/s/##" $R/tools/variants/cw_turn_taxonomy.patch | (cd $V/s && patch -s -p0)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$V/s -I$R/daachorse_amd/csrc -I$R/include -c $V/s/charwise_kernels.hip -o $V/cw.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $V/libdaachorse_amd.so $(ls $R/daachorse_amd/build/*.o | grep -v "/charwise_kernels.hip.o") $V/cw.o
  ls -la $V/libdaachorse_amd.so
else
  python $R/tools/cw_taxonomy.py ${2:-256} 2>&1 | grep -v amdgpu.ids
fi
