#!/bin/bash
# Build timing-experiment variants of the engine (extra -D flags) into .scratch/variants/<name>/libdreamzs.so
# usage: tools/variants.sh name "-DFLAG ..." [name flags]...
cd "$(dirname "$0")/.."
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  mkdir -p gpurun_variants/$n
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Wno-unused-result -w $f -Iinclude -o gpurun_variants/$n/libdreamzs.so pydream_amd/csrc/dz_engine.hip -ldl &
done
wait
ls -la gpurun_variants/*/libdreamzs.so
