#!/bin/bash
exec < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/run_dream_rate.py > gpurun_out/r03h_run_dream_rate.txt 2>&1; tail -2 gpurun_out/r03h_run_dream_rate.txt
python tools/run_dream_rate.py 4096 2000 1 >> gpurun_out/r03h_run_dream_rate.txt 2>&1; tail -1 gpurun_out/r03h_run_dream_rate.txt
python tools/profile_run_dream.py > gpurun_out/r03h_profile_run_dream.txt 2>&1; head -60 gpurun_out/r03h_profile_run_dream.txt
