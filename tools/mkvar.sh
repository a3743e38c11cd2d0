#!/bin/bash
# mkvar.sh NAME FILE.hip -> abtmp/lib_NAME.so: the library with FILE.hip in place of gram_kernels.hip (for tools/ab_libs.sh)
set -e; mkdir -p /tmp/daac_var
R=/root/repo; mkdir -p $R/abtmp
cp $2 /tmp/daac_var/_cur_$1.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/daachorse_amd/csrc -I$R/include -c $2 -o /tmp/daac_var/$1.o 2>/dev/null
objs=$(ls $R/daachorse_amd/build/*.o | grep -v gram_kernels)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/abtmp/lib_$1.so $objs /tmp/daac_var/$1.o 2>/dev/null
ls -la $R/abtmp/lib_$1.so
