#!/bin/bash
# final-tree check: GPU suite, smoke, the driver's bench line, configs[2] with adaptation
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r03j_gputests.log 2>&1
grep -a "passed\|failed\|FAILED" gpurun_out/r03j_gputests.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 20 --warmup 5 > gpurun_out/r03j_bench_k20.json 2> gpurun_out/r03j_bench_k20.err
python bench.py > gpurun_out/r03j_bench.json 2> gpurun_out/r03j_bench.err
python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline > gpurun_out/r03j_bench_mix3_adapt.json 2> gpurun_out/r03j_bench_mix3_adapt.err
python - <<'PY'
import json
for n in ("k20","","mix3_adapt"):
    f="gpurun_out/r03j_bench%s.json" % ("_"+n if n else "")
    d=json.load(open(f)); print(f, "value %.1f" % (d["value"]/1e6), d.get("burnin_value"), d["roofline"]["frac"], d.get("kernel_variant"))
PY
