#!/bin/bash
# Kernel timeline of the bench with its frames in flight (rocprofv3 --kernel-trace does not serialise kernels):
# bash tools/inflight_trace.sh <tag> ; then python3 tools/inflight_trace.py gpurun_out/trace_<tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_${1:-x}
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --skip-cpu --steps 60 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
