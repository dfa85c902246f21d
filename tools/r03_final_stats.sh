#!/bin/bash
# rocprofv3 kernel statistics of the driver's bench command on the final tree
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_stats -o s -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > gpurun_out/final_stats_bench.json 2> gpurun_out/final_stats.err
python - <<'PY'
import csv,glob,json
d=json.load(open("gpurun_out/final_stats_bench.json")); print("bench under rocprof: %.1f M/s, launch_us %.1f (%s)" % (d["value"]/1e6, d["roofline"]["launch_us"], d["roofline"]["launch_us_from"][:40]))
for f in glob.glob("gpurun_out/final_stats/**/s_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-70s %6s %9.1f us %6.2f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
