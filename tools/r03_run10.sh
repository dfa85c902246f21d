#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in xfw1 cur; do
  lib=gpurun_variants/$v/libdreamzs.so; [ $v = cur ] && lib=pydream_amd/libdreamzs.so
  for n in 1024 2048; do
  DREAMZS_LIB=$PWD/$lib python bench.py --chains-per-gpu $n --steps 1000 --warmup 100 --no-cpu-baseline --no-dense --no-events --rhat-max-generations 2000 --rhat-window 500 > gpurun_out/r03j_${v}_${n}_$rep.json 2>/dev/null
  python - gpurun_out/r03j_${v}_${n}_$rep.json $v $n <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2].ljust(6), sys.argv[3], "chains K=1000 %.1f M/s" % (d["value"]/1e6), d["kernel_variant"])
except Exception as ex: print(sys.argv[2], "ERR",ex)
PY
  done
done; done
DZ_BENCH_DEVICE=0 python tools/launch_ranks.py 2 bench.py --gpus 2 --steps 20 --warmup 5 --chains-per-gpu 2048 --no-cpu-baseline --transport peer > gpurun_out/r03j_bench_two_ranks_one_gpu_peer.json 2> gpurun_out/r03j_bench_two_ranks_one_gpu_peer.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03j_bench_two_ranks_one_gpu_peer.json")); print({k:d.get(k) for k in ("value","n_gpus","transport","history_lag","exchange_exposed_us_per_cycle")}, d.get("exchange"))
PY
