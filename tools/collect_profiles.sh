#!/bin/bash
# Runs on the GPU box (through gpurun): the evidence bundle for profiles/.
#   1. rocprofv3 --kernel-trace --stats of the default bench command        -> gpurun_out/<tag>_stats/
#   2. one rocprofv3 --pmc pass per counter group (SQ groups, FETCH_SIZE, WRITE_SIZE), never combined with tracing
#      domains other than --kernel-trace                                    -> gpurun_out/<tag>_pmc_<i>/
#   3. the bench line itself                                                -> gpurun_out/<tag>_bench.json
# usage: tools/collect_profiles.sh <tag> [bench args]
exec < /dev/null
tag=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 600 python bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats -o s --output-format csv -- python bench.py --no-cpu-baseline --no-dense --no-lag0 --no-configs "$@" > gpurun_out/${tag}_stats.log 2>&1
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $group -d gpurun_out/${tag}_pmc_$i -o p --output-format csv -- python bench.py --steps 40 --warmup 10 --spinup 200 --min-timed-ms 5 --no-dense --no-lag0 --no-cpu-baseline --no-events --no-configs "$@" > gpurun_out/${tag}_pmc_$i.log 2>&1
  f=$(find gpurun_out/${tag}_pmc_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f gpurun_out/${tag}_pmc_$i.json > /dev/null
done <<'G'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS
FETCH_SIZE
WRITE_SIZE
G
python - <<PY
import json,glob
out={}
for f in sorted(glob.glob("gpurun_out/${tag}_pmc_*.json")):
    for k,v in json.load(open(f)).items():
        out.setdefault(k,{}).update(v)
json.dump(out,open("gpurun_out/${tag}_pmc_summary.json","w"),indent=1)
for k,v in out.items():
    print(k,{c:round(x) for c,x in v.items()})
PY
# generations of a PMC pass = spin-up (extended by time on one GPU) + warm-up + steps, read back from its bench line
gens=$(python - <<PY
import json,re
line=[l for l in open("gpurun_out/${tag}_pmc_3.log") if l.startswith('{"metric"')][-1]
d=json.loads(line); print(d["generations_executed"])
PY
)
python tools/traffic_from_pmc.py gpurun_out/${tag}_pmc_summary.json $gens gpurun_out/${tag}_traffic.json
find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -12
python tools/benchline.py < gpurun_out/${tag}_bench.json
