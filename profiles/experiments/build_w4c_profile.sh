#!/bin/bash
# Builds build_exp/libmopa_w4cprof.so = the shipped library with wino2d.hip compiled -DW4C_PROFILE: k_wino4_conv prints the in-kernel cycle
# counters of one wave per launch (patch wait, transform, multiply, fold + store).  Numbers in profiles/r4_wino4_one_kernel.md.
#   MOPA_HIP_LIB=$PWD/build_exp/libmopa_w4cprof.so python profiles/bench_wino_direct.py 16
set -e
cd "$(dirname "$0")/../.."
R=$PWD; C=$R/mopa_amd/csrc; O=$R/build_exp; mkdir -p $O
make -C $C -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -DW4C_PROFILE -c $C/wino2d.hip -o $O/wino2d_prof.o
OBJS=$(ls $C/*.o | grep -v '/wino2d.o')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $O/wino2d_prof.o -o $O/libmopa_w4cprof.so
ls -la $O/libmopa_w4cprof.so
