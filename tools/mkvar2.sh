#!/bin/bash
# mkvar2.sh NAME [-DFLAG=..]... -> abtmp/lib_NAME.so: the library with ${SRC:-gram2_kernels}.hip compiled with the given flags
# (for tools/ab_libs3.sh, tools/ab_libs_find.sh, tools/ab_libs_pfx.sh, tools/ab_emit.sh).  Needs an up-to-date daachorse_amd/build/ (python daachorse_amd/_build.py).
# The timing-only modes (G3X_NO_* of gram3_kernels.hip, E3X_NO_* of emit3_kernels.hip: one stage cut out, WRONG results on purpose) are not
# in the shipped sources: tools/variants/<file>_decomposition.patch adds them to a copy of the file before it is compiled here.
set -e; mkdir -p /tmp/daac_var
R=/root/repo; mkdir -p $R/abtmp
N=$1; shift
S=${SRC:-gram2_kernels}
cp $R/daachorse_amd/csrc/$S.hip /tmp/daac_var/$S.hip
P=$R/tools/variants/$(echo $S | sed 's/_kernels//')_decomposition.patch
if [ -f "$P" ]; then (cd /tmp/daac_var && patch -s -p3 $S.hip < $P); fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/daachorse_amd/csrc -I$R/include "$@" -c /tmp/daac_var/$S.hip -o /tmp/daac_var/$N.o 2>/dev/null
objs=$(ls $R/daachorse_amd/build/*.o | grep -v "/$S.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/abtmp/lib_$N.so $objs /tmp/daac_var/$N.o 2>/dev/null
ls -la $R/abtmp/lib_$N.so
