#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r03e_gputests.log 2>&1
grep -a "passed\|failed\|FAILED" gpurun_out/r03e_gputests.log | tail -8
python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline > gpurun_out/r03e_bench_mix3_adapt.json 2> gpurun_out/r03e_bench_mix3_adapt.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > gpurun_out/r03e_bench_k20.json 2> gpurun_out/r03e_bench_k20.err
python tools/variant_rates.py > gpurun_out/r03e_variant_rates.txt 2>&1; cat gpurun_out/r03e_variant_rates.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r03e_stats_mix3_adapt -o s --output-format csv -- python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline --no-events --rhat-max-generations 1000 --rhat-min-generations 500 --rhat-window 200 --min-timed-ms 10 > gpurun_out/r03e_stats_mix3_adapt.log 2>&1
python - <<'PY'
import csv,glob
for r in csv.DictReader(open(glob.glob('gpurun_out/r03e_stats_mix3_adapt/*kernel_stats.csv')[0])):
    print(r['Name'][:40].ljust(40), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), 'us  min', '%.1f'%(float(r['MinNs'])/1e3), r['Percentage'])
PY
for f in gpurun_out/r03e_bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d.get(k) for k in ("value","burnin_value","dense_value","ms_per_step","kernel_variant","rhat_max")}, d.get("roofline",{}).get("frac"))
except Exception as ex: print("ERR",ex)
PY
done
