#!/bin/bash
# large-d parity tests with a variant library: tools/c5_parity.sh <variant>
cp pydream_amd/libdreamzs.so /tmp/libdreamzs.keep2
cp gpurun_variants/$1/libdreamzs.so pydream_amd/libdreamzs.so
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "1000 or big or large or mvn or C5 or c5 or shard" 2>&1 | tail -3
cp /tmp/libdreamzs.keep2 pydream_amd/libdreamzs.so
