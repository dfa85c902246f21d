#!/bin/bash
# quick A/B of experiment builds of k_generations_w4 at BASELINE configs[1] (1024 chains): $1 = library with the experiment (product flags), $2 = instrumented library
exec < /dev/null
cd "$(dirname "$0")/.."
run() { DREAMZS_LIB=$1 DZ_MEGA_W4=$2 python bench.py --chains-per-gpu 1024 --steps $3 --warmup 50 --no-cpu-baseline --no-dense --no-lag0 --no-events --rhat-max-generations 2000 --rhat-window 1000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M/s  %.2f us/gen  %s' % (d['value']/1e6, 1e3*d['ms_per_step'], d['kernel_variant']))"; }
for rep in 1 2; do
  echo -n "experiment K=20:   "; run $PWD/$1 1 20
  echo -n "experiment K=1000: "; run $PWD/$1 1 1000
  echo -n "old kernel K=20:   "; run $PWD/$1 0 20
done
if [ -n "$2" ]; then
  DREAMZS_LIB=$PWD/$2 python bench.py --chains-per-gpu 1024 --steps 200 --warmup 50 --no-cpu-baseline --no-dense --no-lag0 --no-events --rhat-max-generations 1000 --rhat-min-generations 500 --rhat-window 500 > /dev/null 2>&1
  python tools/stamps_w4.py gpurun_out/stamps.bin
fi
