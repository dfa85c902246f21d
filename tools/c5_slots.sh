#!/bin/bash
# k_logp_mvn_gemm resident blocks per CU (DZ_LOGP_SLOTS) x tile (DZ_LOGP_BM), 512 chains x 1000-D
for cfg in "0 32" "3 32" "4 32" "5 32" "6 32" "0 64" "2 64" "3 64" "0 32"; do
  set -- $cfg
  DZ_LOGP_SLOTS=$1 DZ_LOGP_BM=$2 timeout 300 python bench.py --chains-per-gpu 512 --dim 1000 --steps 30 --warmup 10 --rhat-chunk 50 \
      --rhat-max-generations 50 --rhat-min-generations 50 --rhat-window 50 --no-cpu-baseline --no-dense > /tmp/o.json 2> /tmp/o.err
  python - <<PY
import json
try:
    d = json.loads(open("/tmp/o.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("slots $1 bm $2:", round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"] * 1e3, 1), "us/gen; logp launch", round(r["avg_launch_us"], 1), "us", round(r["achieved"], 1), r["unit"])
except Exception as ex:
    print("slots $1 bm $2: failed", ex, open("/tmp/o.err").read()[-300:])
PY
done
