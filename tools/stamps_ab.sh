#!/bin/bash
# cycle stamps of the persistent kernel's last generation for instrumented builds: tools/stamps_ab.sh <variant>...
cp pydream_amd/libdreamzs.so /tmp/libdreamzs.head
for v in "$@"; do
  cp gpurun_variants/$v/libdreamzs.so pydream_amd/libdreamzs.so
  python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-dense --no-events --rhat-max-generations 1000 --rhat-min-generations 500 --rhat-window 500 > /dev/null 2>&1
  echo "== $v"; python tools/stamps_blocks.py gpurun_out/stamps.bin
done
cp /tmp/libdreamzs.head pydream_amd/libdreamzs.so
