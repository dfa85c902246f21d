#!/bin/bash
# configs[4] shard: parity of the large-d cases, then the bench line with / without the k_q_finish launches and the fused accept + propose
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "large_d or d1000 or c5_shard" 2>&1 | tail -5
for v in "0 0" "1 0" "1 1"; do
  set -- $v
  DZ_QFIN=$1 DZ_FUSE_STREAM=$2 python bench.py --chains-per-gpu 512 --dim 1000 --steps 50 --warmup 10 --no-cpu-baseline --no-dense --rhat-max-generations 2000 > gpurun_out/c4b_q$1f$2.json 2> gpurun_out/c4b_q$1f$2.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c4b_q?f?.json")):
    try:
        d=json.load(open(f)); kt=d["kernel_times"]
        print(f, "%.2f M/s  %.1f us/gen" % (d["value"]/1e6, d["ms_per_step"]*1e3), {k:(round(v["avg_us"],1) if v["avg_us"] else None) for k,v in kt.items() if isinstance(v,dict)})
    except Exception as ex: print(f, "ERR", ex)
PY
