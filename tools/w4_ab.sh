#!/bin/bash
# A/B of the small-population kernel (k_generations_w4: the tries' base-independent halves made ahead) against k_generations<.., 4, 4, lean>
# on one box: BASELINE configs[1] (1024 chains) through bench.py at the driver's K = 20 and at K = 1000, alternating.
exec < /dev/null
cd "$(dirname "$0")/.."
for rep in 1 2; do for w in 1 0; do for K in 20 1000; do
  echo -n "DZ_MEGA_W4=$w K=$K: "
  DZ_MEGA_W4=$w python bench.py --chains-per-gpu 1024 --steps $K --warmup 50 --no-cpu-baseline --no-dense --no-lag0 --no-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M/s  %.2f us/gen  %s' % (d['value']/1e6, 1e3*d['ms_per_step'], d['kernel_variant']))"
done; done; done
