#!/bin/bash
# mkvar2.sh NAME [-DFLAG=..]... -> abtmp/lib_NAME.so: the library with ${SRC:-gram2_kernels}.hip compiled with the given flags
# (for tools/ab_libs3.sh, tools/ab_libs_find.sh, tools/ab_libs_pfx.sh, tools/ab_emit.sh).  Needs an up-to-date daachorse_amd/build/ (python daachorse_amd/_build.py).
set -e; mkdir -p /tmp/daac_var
R=/root/repo; mkdir -p $R/abtmp
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/daachorse_amd/csrc -I$R/include "$@" -c $R/daachorse_amd/csrc/${SRC:-gram2_kernels}.hip -o /tmp/daac_var/$N.o 2>/dev/null
objs=$(ls $R/daachorse_amd/build/*.o | grep -v "/${SRC:-gram2_kernels}.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/abtmp/lib_$N.so $objs /tmp/daac_var/$N.o 2>/dev/null
ls -la $R/abtmp/lib_$N.so
