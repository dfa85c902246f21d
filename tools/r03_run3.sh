#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q -x --durations=8 ) > gpurun_out/r03c_gputests.log 2>&1
tail -4 gpurun_out/r03c_gputests.log
for rep in 1 2; do
for v in base cur; do
  lib=gpurun_variants/$v/libdreamzs.so; [ $v = cur ] && lib=pydream_amd/libdreamzs.so
  DREAMZS_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > gpurun_out/r03c_ab_${v}_$rep.json 2> gpurun_out/r03c_ab_${v}_$rep.err
  python - gpurun_out/r03c_ab_${v}_$rep.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]/1e6,1), "M/s  k20;  launch_us", d["roofline"]["launch_us"], d["roofline"].get("launch_us_event_median"))
except Exception as ex: print("ERR",ex)
PY
done; done
DREAMZS_LIB=$PWD/gpurun_variants/base/libdreamzs.so python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-dense --no-events > gpurun_out/r03c_ab_base_k1000.json 2>/dev/null
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-dense --no-events > gpurun_out/r03c_ab_cur_k1000.json 2>/dev/null
python - <<'PY'
import json
for v in ("base","cur"):
    try: print(v, "K=1000", round(json.load(open("gpurun_out/r03c_ab_%s_k1000.json"%v))["value"]/1e6,1))
    except Exception as ex: print(v,"ERR",ex)
PY
python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline > gpurun_out/r03c_bench_mix3_adapt.json 2> gpurun_out/r03c_bench_mix3_adapt.err
python bench.py --steps 20 --warmup 5 --adapt --no-cpu-baseline --no-dense --rhat-max-generations 2000 --rhat-window 500 > gpurun_out/r03c_bench_mvn_adapt.json 2> gpurun_out/r03c_bench_mvn_adapt.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r03c_stats_mix3_adapt -o s --output-format csv -- python bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline --no-events --rhat-max-generations 1000 --rhat-min-generations 500 --rhat-window 200 --min-timed-ms 10 > gpurun_out/r03c_stats_mix3_adapt.log 2>&1
find gpurun_out/r03c_stats_mix3_adapt -name '*kernel_stats.csv' | head -1 | xargs -r cut -c1-150 | head -12
for f in gpurun_out/r03c_bench_*adapt*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d.get(k) for k in ("value","burnin_value","ms_per_step","kernel_variant","rhat_max")}, d.get("burnin"))
except Exception as ex: print("ERR",ex)
PY
done
