#!/bin/bash
# VALU / SALU / LDS instruction counts of the default splat kernel on the bench workload: bash tools/pmc_one.sh [tag]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_one_${1:-x}
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -- python3 bench.py --frames 16 --steps 8 --warmup 2 --skip-cpu --streams 1 > $OUT/log.txt 2>&1
python3 tools/pmc_summary.py $OUT | grep splat
