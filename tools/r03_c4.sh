#!/bin/bash
# configs[4] shard (512 chains x 1000-D): baseline line + chain-group streams
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 1 2 4; do
  DZ_STREAMS=$s python bench.py --chains-per-gpu 512 --dim 1000 --steps 50 --warmup 10 --no-cpu-baseline --no-dense --rhat-max-generations 2000 > gpurun_out/c4_s$s.json 2> gpurun_out/c4_s$s.err
done
for bm in 32 64; do
  DZ_STREAMS=2 DZ_LOGP_BM=$bm python bench.py --chains-per-gpu 512 --dim 1000 --steps 50 --warmup 10 --no-cpu-baseline --no-dense --rhat-max-generations 2000 > gpurun_out/c4_s2_bm$bm.json 2> gpurun_out/c4_s2_bm$bm.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c4_s*.json")):
    try:
        d=json.load(open(f)); kt=d["kernel_times"]
        print(f, "%.2f M/s  %.1f us/gen" % (d["value"]/1e6, d["ms_per_step"]*1e3), {k:(round(v["avg_us"],1) if v["avg_us"] else None) for k,v in kt.items() if isinstance(v,dict)})
    except Exception as ex: print(f, "ERR", ex)
PY
