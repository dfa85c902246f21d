#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 1024 4096; do
DREAMZS_LIB=$PWD/gpurun_variants/stamps/libdreamzs.so python bench.py --chains-per-gpu $n --steps 200 --warmup 50 --no-cpu-baseline --no-dense --no-events --rhat-max-generations 1000 --rhat-min-generations 500 --rhat-window 500 > gpurun_out/r03g_stamps_$n.json 2> gpurun_out/r03g_stamps_$n.err
echo "== $n chains"; python tools/stamps.py gpurun_out/stamps.bin | tail -20; cp gpurun_out/stamps.bin gpurun_out/r03g_stamps_$n.bin
done
