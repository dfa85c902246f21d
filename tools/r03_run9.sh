#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r03i_gputests.log 2>&1
grep -a "passed\|failed\|FAILED" gpurun_out/r03i_gputests.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline > gpurun_out/r03i_bench_mix3_adapt.json 2> gpurun_out/r03i_bench_mix3_adapt.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03i_bench_mix3_adapt.json")); print("mix3 adapt: value %.1f burnin %.1f" % (d["value"]/1e6, d["burnin_value"]/1e6))
PY
