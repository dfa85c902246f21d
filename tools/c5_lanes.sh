#!/bin/bash
# configs[4] per-GPU shard (512 chains x 1000-D) with 1..4 chain-group streams
mkdir -p gpurun_out/c5
for s in 1 2 3 4; do
  DZ_STREAMS=$s timeout 300 python bench.py --chains-per-gpu 512 --dim 1000 --steps 50 --warmup 10 --rhat-chunk 100 \
      --rhat-max-generations 400 --rhat-min-generations 100 --rhat-window 100 --no-cpu-baseline --no-dense > gpurun_out/c5/lanes$s.json 2> gpurun_out/c5/lanes$s.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/c5/lanes$s.json").read().strip().splitlines()[-1])
    print("streams $s:", round(d["value"] / 1e6, 2), "M/s", d["ms_per_step"], "ms/gen")
except Exception as ex:
    print("streams $s: failed", ex)
PY
done
