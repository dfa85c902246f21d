#!/bin/bash
# configs[4] per-GPU shard (512 chains x 1000-D) for every engine build under gpurun_variants/
mkdir -p gpurun_out/c5v
cp pydream_amd/libdreamzs.so /tmp/libdreamzs.keep
for v in "$@"; do
  cp gpurun_variants/$v/libdreamzs.so pydream_amd/libdreamzs.so
  timeout 300 python bench.py --chains-per-gpu 512 --dim 1000 --steps 50 --warmup 10 --rhat-chunk 100 \
      --rhat-max-generations 200 --rhat-min-generations 100 --rhat-window 100 --no-cpu-baseline --no-dense > gpurun_out/c5v/$v.json 2> gpurun_out/c5v/$v.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c5v/$v.json").read().strip().splitlines()[-1])
    print("$v:", round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"] * 1e3, 1), "us/gen; logp launch", round(d["roofline"]["avg_launch_us"], 1), "us")
except Exception as ex:
    print("$v: failed", ex)
PY
done
cp /tmp/libdreamzs.keep pydream_amd/libdreamzs.so
