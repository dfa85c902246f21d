#!/bin/bash
# usage: tools/pmc_c5.sh <outprefix>   (GPU box; one rocprofv3 --pmc pass per counter group on the configs[4] shard run)
exec < /dev/null
pre=$1; shift
export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $group -d gpurun_out/${pre}_$i -o p --output-format csv -- python bench.py --chains-per-gpu 512 --dim 1000 --steps 10 --warmup 2 --rhat-chunk 20 --rhat-max-generations 20 --rhat-min-generations 20 --rhat-window 20 --no-cpu-baseline --no-dense --no-events "$@" > gpurun_out/${pre}_$i.log 2>&1
  f=$(find gpurun_out/${pre}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f gpurun_out/${pre}_$i.json > /dev/null || tail -3 gpurun_out/${pre}_$i.log
  rm -rf gpurun_out/${pre}_$i
done <<'G'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY
SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
GRBM_GUI_ACTIVE TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
G
python - <<PY
import json,glob
out={}
for f in sorted(glob.glob("gpurun_out/${pre}_*.json")):
    if f.endswith("summary.json"): continue
    for k,v in json.load(open(f)).items():
        out.setdefault(k,{}).update(v)
json.dump(out,open("gpurun_out/${pre}_summary.json","w"),indent=1)
for k,v in out.items():
    if "gemm" in k or "stream" in k: print(k,{c:round(x) for c,x in v.items()})
PY
