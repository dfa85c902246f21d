#!/bin/bash
# Measurement builds of the headline kernel only: the NRT = 7 translation unit compiled with extra -D flags, linked with the product's
# other objects (pydream_amd/build/*.o must be current: python -m pydream_amd.build first).
# usage: tools/variant_nrt7.sh name "-DFLAG ..." [name flags]...
cd "$(dirname "$0")/.."
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  mkdir -p gpurun_variants/$n /tmp/var_$n
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result -w -DDZ_TU_NRT=7 $f -c pydream_amd/csrc/dz_mega_tu.hip -o /tmp/var_$n/dz_mega_nrt7.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/$n/libdreamzs.so pydream_amd/build/dz_engine.o $(ls pydream_amd/build/dz_mega_nrt[1-68].o) /tmp/var_$n/dz_mega_nrt7.o -ldl ) &
done
wait
ls -la gpurun_variants/*/libdreamzs.so
