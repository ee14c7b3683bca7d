#!/bin/bash
# Profile the default bench.py workload on the GPU box with rocprofv3:
#   pass 1: --kernel-trace --stats            -> profiles/<tag>_kernel_stats.csv
#   pass 2: --pmc FETCH_SIZE, pass 3: --pmc WRITE_SIZE (separate runs, TCC slots)
#   -> profiles/traffic.json (read by bench.py for roofline.traffic)
# (a bench step is one move of every game = 800 rounds: one launch of the persistent pair in the headline leg, 800
#  launches per kernel in the lockstep leg that follows it; the counter passes run the lockstep form only, because
#  counter collection serialises launches and the pair's two kernels must be resident together)
# usage (through gpurun):  bash tools/profile_bench.sh r2_final
set -e
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --full-games 0 > $OUT/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --lockstep --no-other-configs --full-games 0 > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --lockstep --no-other-configs --full-games 0 > $OUT/write.log 2>&1
python3 tools/summarize_profile.py $OUT $TAG
