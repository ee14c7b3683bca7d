#!/bin/bash
# SQ counters of the default bench.py workload, three rocprofv3 --pmc passes of eight counters, each a run of its
# own (the hardware holds at most 8 SQ counters per pass), summarised per kernel and per launch into
# gpurun_out/pmc_<tag>/summary.json.   usage (through gpurun):  bash tools/pmc_bench.sh r2
# A second argument replaces the profiled program and its arguments, e.g. "tools/tower_bench.py --iters 2".
set -e
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS=${2:-"bench.py --steps 1 --warmup 0 --no-cpu-baseline --lockstep --no-other-configs --full-games 0"}
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1
python3 tools/summarize_pmc.py $OUT $TAG
