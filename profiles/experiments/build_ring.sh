#!/bin/bash
# Builds build_exp/libmopa_ring.so = the shipped library + the round-4 persistent "ring" sparse-convolution kernel
# (profiles/experiments/spconv_ring.hip) behind MOPA_SPCONV_RING=1|2, and the -DRG_PROFILE / -DRG_NOWAIT -DRG_NOLOAD probe variants.
#   MOPA_HIP_LIB=$PWD/build_exp/libmopa_ring.so MOPA_SPCONV_RING=2 python profiles/bench_spconv.py 7 10 8
set -e
cd "$(dirname "$0")/../.."
R=$PWD; E=$R/profiles/experiments; C=$R/mopa_amd/csrc; O=$R/build_exp; mkdir -p $O
make -C $C -j8 > /dev/null
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -I$C -I$E"
/opt/rocm/bin/hipcc $F -DMOPA_EXP_RING -c $C/spconv.hip -o $O/spconv_exp.o
OBJS=$(ls $C/*.o | grep -v '/spconv.o')
for v in ring:"" ringprof:"-DRG_PROFILE" ringnoload:"-DRG_NOWAIT -DRG_NOLOAD"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $F $flags -c $E/spconv_ring.hip -o $O/$name.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $O/spconv_exp.o $O/$name.o -o $O/libmopa_$name.so
done
ls -la $O/*.so
