#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/r03_knockouts.sh ko_all_3 ko_all_3ns 2>&1 | grep -v "^cur"
for v in ko_all_3ns; do
i=0
for group in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  DREAMZS_LIB=$PWD/gpurun_variants/$v/libdreamzs.so timeout 300 rocprofv3 --kernel-trace --pmc $group -d gpurun_out/ko_pmc_${v}_$i -o p --output-format csv -- python bench.py --steps 40 --warmup 10 --spinup 200 --min-timed-ms 5 --no-dense --no-cpu-baseline --no-events > gpurun_out/ko_pmc_${v}_$i.log 2>&1
  f=$(find gpurun_out/ko_pmc_${v}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f gpurun_out/ko_pmc_${v}_$i.json | grep k_generations
done
done
