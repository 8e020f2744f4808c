#!/bin/bash
# build_exp/libmopa_c9prof.so = the shipped library with wino4c9.hip compiled -DC9_PROFILE: k_wino4_conv9 prints the in-kernel cycle
# counters of one wave per launch.   MOPA_HIP_LIB=$PWD/build_exp/libmopa_c9prof.so python profiles/bench_conv9.py 16 1
set -e
cd "$(dirname "$0")/../.."
R=$PWD; C=$R/mopa_amd/csrc; O=$R/build_exp; mkdir -p $O
make -C $C -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -DC9_PROFILE -c $C/wino4c9.hip -o $O/wino4c9_prof.o
OBJS=$(ls $C/*.o | grep -v '/wino4c9.o')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $O/wino4c9_prof.o -o $O/libmopa_c9prof.so
ls -la $O/libmopa_c9prof.so
