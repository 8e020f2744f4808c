#!/bin/bash
# Builds build_exp/libmopa_wgr{1,2,3}.so = the shipped library with sprun.hip compiled -DWGR_PROBE=n: k_wgrad_run without its MFMAs (1),
# without its row gathers (2), with neither (3) -- where does the kernel's time go.  MOPA_HIP_LIB=$PWD/build_exp/libmopa_wgr1.so
# MOPA_SPCONV_WGRAD_RUN=2 python profiles/bench_wgrad.py 5 10   (results are wrong by construction; only the `run us` column means anything)
set -e
cd "$(dirname "$0")/../.."
R=$PWD; C=$R/mopa_amd/csrc; O=$R/build_exp; mkdir -p $O
make -C $C -j8 > /dev/null
for n in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -DWGR_PROBE=$n -c $C/sprun.hip -o $O/sprun_probe$n.o
  OBJS=$(ls $C/*.o | grep -v '/sprun.o')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $O/sprun_probe$n.o -o $O/libmopa_wgr$n.so
done
ls -la $O/libmopa_wgr*.so
