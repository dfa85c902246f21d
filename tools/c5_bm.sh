#!/bin/bash
# k_logp_mvn_gemm tile sizes (DZ_LOGP_BM) at two chain counts, 1000-D
for n in 512 4096; do
for bm in 32 64 128; do
  DZ_LOGP_BM=$bm timeout 300 python bench.py --chains-per-gpu $n --dim 1000 --steps 30 --warmup 10 --rhat-chunk 50 \
      --rhat-max-generations 50 --rhat-min-generations 50 --rhat-window 50 --no-cpu-baseline --no-dense > /tmp/o.json 2> /tmp/o.err
  python - <<PY
import json
try:
    d = json.loads(open("/tmp/o.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("chains $n bm $bm:", round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"] * 1e3, 1), "us/gen; logp launch", round(r["avg_launch_us"], 1), "us", round(r["achieved"], 1), r["unit"])
except Exception as ex:
    print("chains $n bm $bm: failed", ex, open("/tmp/o.err").read()[-300:])
PY
done; done
