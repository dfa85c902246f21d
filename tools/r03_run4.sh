#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -x --durations=5 ) > gpurun_out/r03d_gputests.log 2>&1
tail -3 gpurun_out/r03d_gputests.log
python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline > gpurun_out/r03d_bench_mix3_adapt.json 2> gpurun_out/r03d_bench_mix3_adapt.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r03d_stats_mix3_adapt -o s --output-format csv -- python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline --no-events --rhat-max-generations 1000 --rhat-min-generations 500 --rhat-window 200 --min-timed-ms 10 > gpurun_out/r03d_stats_mix3_adapt.log 2>&1
find gpurun_out/r03d_stats_mix3_adapt -name '*kernel_stats.csv' | head -1 | xargs -r cut -c1-60,150-260 | head -8
python tools/run_dream_rate.py > gpurun_out/r03d_run_dream_rate.txt 2>&1; tail -3 gpurun_out/r03d_run_dream_rate.txt
python bench.py --steps 20 --warmup 5 --chains-per-gpu 1024 --no-cpu-baseline > gpurun_out/r03d_bench_c1.json 2> gpurun_out/r03d_bench_c1.err
for f in gpurun_out/r03d_bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d.get(k) for k in ("value","burnin_value","dense_value","ms_per_step","kernel_variant","rhat_max")}, d.get("roofline",{}).get("frac"))
except Exception as ex: print("ERR",ex)
PY
done
