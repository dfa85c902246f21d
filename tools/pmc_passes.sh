#!/bin/bash
# usage: tools/pmc_passes.sh <outprefix> <bench args...>   (runs on the GPU box; one rocprofv3 --pmc pass per counter group)
exec < /dev/null
pre=$1; shift
export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $group -d gpurun_out/${pre}_$i -o p --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-events "$@" > gpurun_out/${pre}_$i.log 2>&1
  f=$(find gpurun_out/${pre}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f gpurun_out/${pre}_$i.json > /dev/null
done <<'G'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_IFETCH
G
python - <<PY
import json,glob
out={}
for f in sorted(glob.glob("gpurun_out/${pre}_*.json")):
    for k,v in json.load(open(f)).items():
        out.setdefault(k,{}).update(v)
json.dump(out,open("gpurun_out/${pre}_summary.json","w"),indent=1)
for k,v in out.items():
    if "propose" in k or "accept" in k or "logp" in k: print(k,{c:round(x) for c,x in v.items()})
PY
