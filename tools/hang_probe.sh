#!/bin/bash
# repeat the multi-process GPU tests with stack dumps on a stall
for i in $(seq 1 ${1:-6}); do
  t0=$(date +%s)
  timeout 330 python -X faulthandler -m pytest tests/test_api_gpu.py tests/test_distributed.py -m gpu -q -p no:cacheprovider -x -o faulthandler_timeout=120 > gpurun_out/hang_$i.log 2>&1
  echo "run $i rc=$? $(( $(date +%s) - t0 )) s: $(grep -E 'passed|failed' gpurun_out/hang_$i.log | tail -1)"
done
