#!/bin/bash
# full-code (priors / boundaries / several pairs / redraw) instantiations: parity, then their rates
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_api_gpu.py -m gpu -q -k "persistent or impossible or prior or redraw or pairs" 2>&1 | grep -a "passed\|failed\|FAILED" | tail -5
python tools/variant_rates.py 2>&1 | tee gpurun_out/r03_variant_rates_pb.txt
