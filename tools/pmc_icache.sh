#!/bin/bash
# Instruction-cache counters of the splat kernel, on the accumulated-cloud frame (tools/tile_hist.py) and on the bench workload.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}
CTR="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"
OUT=gpurun_out/pmc_icache_${TAG}_heavy
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc $CTR --output-format csv -d $OUT -- python3 tools/tile_hist.py 64 > $OUT/log.txt 2>&1
python3 tools/pmc_summary.py $OUT | grep -E "splat" | cut -c1-400
OUT=gpurun_out/pmc_icache_${TAG}_light
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc $CTR --output-format csv -d $OUT -- python3 bench.py --frames 16 --steps 8 --warmup 2 --skip-cpu --streams 1 > $OUT/log.txt 2>&1
python3 tools/pmc_summary.py $OUT | grep -E "splat" | cut -c1-400
