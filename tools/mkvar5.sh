#!/bin/bash
# mkvar5.sh NAME SRC [-DFLAG=..]... -> abtmp/lib_NAME.so: the library with daachorse_amd/csrc/SRC.hip compiled with the given flags (timing
# experiments; several may run side by side).  Needs an up-to-date daachorse_amd/build/ (python daachorse_amd/_build.py).
set -e
R=/root/repo; mkdir -p $R/abtmp /tmp/daac_var5
N=$1; S=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/daachorse_amd/csrc -I$R/include "$@" -c $R/daachorse_amd/csrc/$S.hip -o /tmp/daac_var5/$N.o
objs=$(ls $R/daachorse_amd/build/*.o | grep -v "/$S.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/abtmp/lib_$N.so $objs /tmp/daac_var5/$N.o
ls -la $R/abtmp/lib_$N.so
