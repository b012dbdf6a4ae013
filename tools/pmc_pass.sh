#!/bin/bash
# One rocprofv3 counter pass over a short bench run: bash tools/pmc_pass.sh <tag> "<counters>" [bench args]
set -e
TAG=$1; CTR=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc $CTR --output-format csv -d $OUT -- python3 bench.py --frames 16 --steps 8 --warmup 2 --skip-cpu --streams 1 "$@" > $OUT/log.txt 2>&1
python3 tools/pmc_summary.py $OUT
