#!/bin/bash
# Knock-out builds of the headline kernel (tools/variant_nrt7.sh with -DDZ_KO_*: one part of the generation removed, results wrong on
# purpose): what each part costs IN the pipeline, measured on one box against the product build.
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in cur "$@" cur; do
  lib=gpurun_variants/$v/libdreamzs.so; [ $v = cur ] && lib=pydream_amd/libdreamzs.so
  DREAMZS_LIB=$PWD/$lib python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-dense --no-events --rhat-max-generations 3000 --rhat-min-generations 3000 --rhat-window 500 > gpurun_out/ko_$v.json 2>gpurun_out/ko_$v.err
  python - gpurun_out/ko_$v.json $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2].ljust(12), "%.2f us per generation  (%.1f M/s)  acceptance %.3f" % (1e3*d["ms_per_step"], d["value"]/1e6, d["acceptance_rate"]))
except Exception as ex: print(sys.argv[2], "ERR", ex)
PY
done
