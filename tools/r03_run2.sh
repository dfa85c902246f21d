#!/bin/bash
# configs[2] with adaptation on: new burn-in path vs the multi-kernel one; kernel stats of the burn-in
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline > gpurun_out/r03b_bench_mix3_adapt.json 2> gpurun_out/r03b_bench_mix3_adapt.err
DZ_MEGA_BURNIN=0 python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline --rhat-max-generations 2000 --rhat-window 500 > gpurun_out/r03b_bench_mix3_adapt_oldpath.json 2> gpurun_out/r03b_bench_mix3_adapt_oldpath.err
python bench.py --steps 20 --warmup 5 --adapt --no-cpu-baseline --no-dense --rhat-max-generations 2000 --rhat-window 500 > gpurun_out/r03b_bench_mvn_adapt.json 2> gpurun_out/r03b_bench_mvn_adapt.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r03b_stats_mix3_adapt -o s --output-format csv -- python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline --no-events --rhat-max-generations 1000 --rhat-min-generations 500 --rhat-window 200 --min-timed-ms 10 > gpurun_out/r03b_stats_mix3_adapt.log 2>&1
find gpurun_out/r03b_stats_mix3_adapt -name '*kernel_stats.csv' | head -1 | xargs -r head -14
for f in gpurun_out/r03b_bench_*adapt*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d.get(k) for k in ("value","burnin_value","ms_per_step","kernel_variant","rhat_max")}, d.get("burnin"))
except Exception as ex: print("ERR",ex)
PY
done
