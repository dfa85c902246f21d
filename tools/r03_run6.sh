#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r03f_gputests.log 2>&1
grep -a "passed\|failed\|FAILED" gpurun_out/r03f_gputests.log | tail -8
bash tools/r03_ab.sh r03f nobatch
python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline > gpurun_out/r03f_bench_mix3_adapt.json 2> gpurun_out/r03f_bench_mix3_adapt.err
python tools/variant_rates.py > gpurun_out/r03f_variant_rates.txt 2>&1; cat gpurun_out/r03f_variant_rates.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03f_bench_mix3_adapt.json")); print("mix3 adapt: value %.1f burnin %.1f" % (d["value"]/1e6, d["burnin_value"]/1e6))
PY
