#!/bin/bash
# Several builds of the library on one box, alternating: tools/ab_many.sh K reps lib1.so lib2.so ...   (CHAINS=1024: BASELINE configs[1])
exec < /dev/null
cd "$(dirname "$0")/.."
K=$1; R=$2; shift 2
for rep in $(seq $R); do for lib in "$@"; do
  echo -n "$(basename $lib) K=$K: "
  DREAMZS_LIB=$PWD/$lib python bench.py ${CHAINS:+--chains-per-gpu $CHAINS} ${EXTRA} --steps $K --warmup 50 --no-cpu-baseline --no-dense --no-lag0 --no-events --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M/s  %.2f us/gen  %s' % (d['value']/1e6, 1e3*d['ms_per_step'], d['kernel_variant']))"
done; done
