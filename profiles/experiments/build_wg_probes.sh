#!/bin/bash
# Timing probes of k_wino4_wgrad (csrc/wino4wg.hip): build_exp/libmopa_wg_<probe>.so = the shipped library with that file compiled
# -DWG_PROBE_<X> (wrong results on purpose: each probe removes one component of the loop).
#   MOPA_HIP_LIB=$PWD/build_exp/libmopa_wg_nomfma.so python profiles/bench_wgrad2d.py 16
set -e
cd "$(dirname "$0")/../.."
R=$PWD; C=$R/mopa_amd/csrc; O=$R/build_exp; mkdir -p $O
make -C $C -j8 > /dev/null
OBJS=$(ls $C/*.o | grep -v '/wino4wg.o')
for P in ${PROBES:-NOMFMA NOLOAD NOEPI}; do
  L=$(echo $P | tr A-Z a-z)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -DWG_PROBE_$P -DWG_$P -c $C/wino4wg.hip -o $O/wino4wg_$L.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $O/wino4wg_$L.o -o $O/libmopa_wg_$L.so
done
ls -la $O/libmopa_wg_*.so
