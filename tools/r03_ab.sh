#!/bin/bash
# A/B of library builds on one box: gpurun_variants/<name>/libdreamzs.so vs the in-tree build ("cur"), alternating, headline workload
# usage: tools/r03_ab.sh <tag> <variant> [<variant> ...]
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
for v in "$@" cur; do
  lib=gpurun_variants/$v/libdreamzs.so; [ $v = cur ] && lib=pydream_amd/libdreamzs.so
  DREAMZS_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > gpurun_out/${tag}_ab_${v}_$rep.json 2> gpurun_out/${tag}_ab_${v}_$rep.err
  DREAMZS_LIB=$PWD/$lib python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-dense --no-events --rhat-max-generations 2000 --rhat-window 500 > gpurun_out/${tag}_ab_${v}_k1000_$rep.json 2>/dev/null
  python - gpurun_out/${tag}_ab_${v}_$rep.json gpurun_out/${tag}_ab_${v}_k1000_$rep.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); e=json.load(open(sys.argv[2]))
    print(sys.argv[3].ljust(10), "K=20 %.1f M/s" % (d["value"]/1e6), " K=1000 %.1f M/s" % (e["value"]/1e6), " frac", round(d["roofline"]["frac"],4))
except Exception as ex: print(sys.argv[3], "ERR",ex)
PY
done; done
