#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r03m_gputests.log 2>&1
grep -a "passed\|failed\|FAILED" gpurun_out/r03m_gputests.log | tail -8
for rep in 1 2; do
python bench.py --steps 1000 --warmup 100 --target mix3 --no-cpu-baseline --no-events > gpurun_out/r03m_mix3_$rep.json 2>/dev/null
python - gpurun_out/r03m_mix3_$rep.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("mix3 K=1000 %.1f M/s" % (d["value"]/1e6), d["kernel_variant"])
except Exception as ex: print("ERR",ex)
PY
done
python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline > gpurun_out/r03m_bench_mix3_adapt.json 2> gpurun_out/r03m_bench_mix3_adapt.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03m_bench_mix3_adapt.json")); print("mix3 adapt K=20: value %.1f burnin %.1f" % (d["value"]/1e6, d["burnin_value"]/1e6))
PY
python tools/variant_rates.py "boundaries" 2>&1 | tail -3
