#!/bin/bash
# mkvar5.sh NAME SRC [-DFLAG=..]... -> abtmp/lib_NAME.so: the library with daachorse_amd/csrc/SRC.hip compiled with the given flags (timing
# experiments; several may run side by side).  Needs an up-to-date daachorse_amd/build/ (python daachorse_amd/_build.py).
# The timing-only stage switches of the count kernel (-DG4X=1..5: main path only / + hit queue / + consumer up to the rank (filter body: the filter and the re-compaction; -DG4X=6: + the survivors ranked) / + record gather
# and pending stage / + walker slab without the drain; -DG4X=9: fewer template instances; -DG4_GS=n: lookups in flight; WRONG counts on
# purpose) are not in the shipped source: tools/variants/<SRC minus _kernels>_decomposition.patch adds them to a copy compiled here.
set -e
R=/root/repo; mkdir -p $R/abtmp /tmp/daac_var5/$1
N=$1; S=$2; shift; shift
cp $R/daachorse_amd/csrc/$S.hip /tmp/daac_var5/$N/$S.hip
P=$R/tools/variants/$(echo $S | sed 's/_kernels//')_decomposition.patch
if [ -f "$P" ]; then (cd /tmp/daac_var5/$N && patch -s $S.hip < $P); fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/daachorse_amd/csrc -I$R/include "$@" -c /tmp/daac_var5/$N/$S.hip -o /tmp/daac_var5/$N.o
objs=$(ls $R/daachorse_amd/build/*.o | grep -v "/$S.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/abtmp/lib_$N.so $objs /tmp/daac_var5/$N.o
ls -la $R/abtmp/lib_$N.so
