#!/bin/bash
# configs[4] shard: the full bench line (convergence budget 16 d generations, dense leg, CPU baseline) + kernel stats
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed" | tail -3
python bench.py --chains-per-gpu 512 --dim 1000 --steps 50 --warmup 10 --cpu-chains 512 --cpu-seconds 10 > gpurun_out/r03_bench_config4_1000d_512chains.json 2> gpurun_out/r03_bench_config4.err
tail -3 gpurun_out/r03_bench_config4.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4stats -o s -- python bench.py --chains-per-gpu 512 --dim 1000 --steps 50 --warmup 10 --no-cpu-baseline --no-dense --rhat-max-generations 2000 > /dev/null 2> gpurun_out/c4stats.err
python - <<'PY'
import json,csv,glob
d=json.load(open("gpurun_out/r03_bench_config4_1000d_512chains.json"))
print("value %.2f M/s, %.1f us/gen, rhat_max %s, to<1.2: %s, roofline %s" % (d["value"]/1e6, d["ms_per_step"]*1e3, d.get("rhat_max"), d["convergence"].get("generations_to_rhat_below_1p2"), d["roofline"]))
print(d.get("cpu_baseline"))
for f in glob.glob("gpurun_out/c4stats/**/s_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print("%-60s %6s %9.1f us %6.2f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
