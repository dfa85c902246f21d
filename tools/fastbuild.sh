#!/bin/bash
# Experiment build: only the NRT = 7 translation unit (d = 100) with the 16-chains-per-block multi-try instantiations, linked with the
# regular build's other objects into pydream_amd/build/libdreamzs_fast.so.  Use: DREAMZS_LIB=pydream_amd/libdreamzs_fast.so (OUT=... picks another name) python ...
# usage: tools/fastbuild.sh [extra hipcc flags, e.g. -DDZ_EXP_X]
cd "$(dirname "$0")/.."
B=pydream_amd/build; F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $F -DDZ_TU_NRT=7 -DDZ_TU_FAST "$@" -c pydream_amd/csrc/dz_mega_tu.hip -o $B/fast_nrt7.o &
/opt/rocm/bin/hipcc $F "$@" -c pydream_amd/csrc/dz_engine.hip -o $B/fast_engine.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ${OUT:-pydream_amd/libdreamzs_fast.so} $B/fast_engine.o $B/dz_mega_nrt1.o $B/dz_mega_nrt2.o $B/dz_mega_nrt3.o $B/dz_mega_nrt4.o $B/dz_mega_nrt5.o $B/dz_mega_nrt6.o $B/fast_nrt7.o $B/dz_mega_nrt8.o $B/dz_mega_nrt9.o $B/dz_mega_nrt10.o $B/dz_mega_nrt11.o $B/dz_mega_nrt12.o $B/dz_mega_nrt13.o $B/dz_mega_nrt14.o $B/dz_mega_nrt15.o $B/dz_mega_nrt16.o -ldl && echo built ${OUT:-pydream_amd/libdreamzs_fast.so}
