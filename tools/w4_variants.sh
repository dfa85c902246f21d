#!/bin/bash
# experiment builds of the NRT = 7 translation unit with extra -D flags, linked against the regular objects: tools/w4_variants.sh name "flags" [name "flags"] ...
# -> pydream_amd/build/var_<name>.so
cd "$(dirname "$0")/.."
B=pydream_amd/build; F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result"
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  ( /opt/rocm/bin/hipcc $F -DDZ_TU_NRT=7 -DDZ_TU_FAST $f -c pydream_amd/csrc/dz_mega_tu.hip -o $B/var_$n.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/var_$n.so $B/dz_engine.o $B/dz_mega_nrt1.o $B/dz_mega_nrt2.o $B/dz_mega_nrt3.o $B/dz_mega_nrt4.o $B/dz_mega_nrt5.o $B/dz_mega_nrt6.o $B/var_$n.o $B/dz_mega_nrt8.o $B/dz_mega_nrt9.o $B/dz_mega_nrt10.o $B/dz_mega_nrt11.o $B/dz_mega_nrt12.o $B/dz_mega_nrt13.o $B/dz_mega_nrt14.o $B/dz_mega_nrt15.o $B/dz_mega_nrt16.o -ldl && echo built var_$n ) &
done
wait
