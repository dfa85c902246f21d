#!/bin/bash
# burn-in rates (bench.py --adapt) of the mixture (configs[2]) and MVN targets: value / burnin_value / us per burn-in generation / probabilities
# usage (on the GPU box): tools/adapt_rates.sh [extra bench args]
exec < /dev/null
cd "$(dirname "$0")/.."
for t in "--target mix3" ""; do
  python bench.py --steps 20 --warmup 5 $t --adapt --no-cpu-baseline --no-dense --no-lag0 "$@" 2>/dev/null | python -c '
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); b=d["burnin"]
print(round(d["value"]/1e6,1), round(d["burnin_value"]/1e6,1), round(b["ms_per_step"]*1e3,2), b["kernel_variant"], b["cr_probs_after_burnin"])'
done
