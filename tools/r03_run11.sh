#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r03k_gputests.log 2>&1
grep -a "passed\|failed\|FAILED" gpurun_out/r03k_gputests.log | tail -8
for rep in 1 2; do
  python bench.py --chains-per-gpu 1024 --steps 1000 --warmup 100 --no-cpu-baseline --no-dense --no-events > gpurun_out/r03k_c1_$rep.json 2>/dev/null
  python - gpurun_out/r03k_c1_$rep.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("1024 chains K=1000 %.1f M/s" % (d["value"]/1e6), d["kernel_variant"])
except Exception as ex: print("ERR",ex)
PY
done
python bench.py --chains-per-gpu 1024 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03k_bench_c1.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03k_bench_c1.json")); print("1024 chains K=20 %.1f M/s dense %.1f frac %.3f" % (d["value"]/1e6, d["dense_value"]/1e6, d["roofline"]["frac"]))
PY
