#!/bin/bash
# A/B of two builds of the library on the headline workload through bench.py, alternating: tools/ab_libs.sh <libA.so> <libB.so> [K] [reps]
# (CHAINS=1024 in the environment: BASELINE configs[1] instead of the headline's 4096 chains)
exec < /dev/null
cd "$(dirname "$0")/.."
K=${3:-20}; R=${4:-3}
for rep in $(seq $R); do for lib in "$1" "$2"; do
  echo -n "$(basename $lib) K=$K: "
  DREAMZS_LIB=$PWD/$lib python bench.py ${CHAINS:+--chains-per-gpu $CHAINS} --steps $K --warmup 50 --no-cpu-baseline --no-dense --no-lag0 --no-events --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M/s  %.2f us/gen  %s' % (d['value']/1e6, 1e3*d['ms_per_step'], d['kernel_variant']))"
done; done
