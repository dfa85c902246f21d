#!/bin/bash
# The round's evidence bundle, on the GPU box (through gpurun):  tools/collect_round.sh <tag>
#   1. tools/collect_profiles.sh <tag>: default bench line, rocprofv3 --kernel-trace --stats, the --pmc passes, HBM traffic
#   2. bench lines of the other BASELINE configurations and of the multi-rank forms rehearsed on this one GPU
#   3. rocprofv3 kernel stats of configs[2] with adaptation (the crossover burn-in) and of the configs[4] shard
#   4. variant rates, the examples' convergence loop end to end
exec < /dev/null
tag=$1
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=$(pwd)
tools/collect_profiles.sh $tag > gpurun_out/${tag}_collect.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_k20.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --chains-per-gpu 1024 > gpurun_out/${tag}_bench_config1_1024chains.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --target mix3 --adapt > gpurun_out/${tag}_bench_config2_mixture_adapt.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --adapt --no-dense > gpurun_out/${tag}_bench_mvn_adapt.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --chains-per-gpu 512 --dim 1000 --no-dense > gpurun_out/${tag}_bench_config4_1000d_512chains.json 2> /dev/null
DZ_BENCH_DEVICE=0 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_gpus2_self_launch_one_gpu.json 2> gpurun_out/${tag}_bench_gpus2.err
DZ_BENCH_DEVICE=0 python3 bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --rhat-max-generations 1500 --rhat-min-generations 1000 --rhat-window 500 > gpurun_out/${tag}_bench_eight_ranks_one_gpu_rehearsal.json 2> gpurun_out/${tag}_bench_gpus8.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats_c2 -o s --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --target mix3 --adapt --no-cpu-baseline > /dev/null 2>&1)
cp "$(find gpurun_out/${tag}_stats_c2 -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats_config2_adapt.csv
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats_c4 -o s --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --chains-per-gpu 512 --dim 1000 --no-dense --no-cpu-baseline > /dev/null 2>&1)
cp "$(find gpurun_out/${tag}_stats_c4 -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats_config4_1000d.csv
cp "$(find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_kernel_stats.csv
python tools/variant_rates.py > gpurun_out/${tag}_variant_rates.txt 2>&1
python tools/run_dream_rate.py 2>&1 | grep -v Warning > gpurun_out/${tag}_run_dream_loop.txt
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_stats_c2 gpurun_out/${tag}_stats_c4 gpurun_out/${tag}_pmc_[0-9]
for f in gpurun_out/${tag}_bench*.json; do echo "$f: $(python tools/benchline.py < $f 2>&1 | head -1)"; done
head -6 gpurun_out/${tag}_kernel_stats_config2_adapt.csv | cut -c1-160
