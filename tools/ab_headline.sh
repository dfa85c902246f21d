#!/bin/bash
# A/B of engine builds on the headline workload: tools/ab_headline.sh <variant> [<variant>...]  ("head" = the in-tree library)
cp pydream_amd/libdreamzs.so /tmp/libdreamzs.head
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = head ]; then cp /tmp/libdreamzs.head pydream_amd/libdreamzs.so; else cp gpurun_variants/$v/libdreamzs.so pydream_amd/libdreamzs.so; fi
  echo -n "$v: "; python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-dense ${BENCH_ARGS} 2>/dev/null | python tools/benchline.py 2>/dev/null | tail -1 | cut -c1-110
done; done
cp /tmp/libdreamzs.head pydream_amd/libdreamzs.so
