#!/bin/bash
# Collects the rocprofv3 evidence for one round: kernel-trace stats of the default bench command and two PMC passes
# (HBM read / write bytes), an SQ instruction / wait pass and an LDS pass.  Run on the GPU box: bash tools/profile_round.sh r01
set -e
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
rm -rf $OUT
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --skip-cpu --streams 1 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --frames 16 --steps 8 --warmup 2 --skip-cpu --streams 1 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --frames 16 --steps 8 --warmup 2 --skip-cpu --streams 1 > $OUT/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --frames 16 --steps 8 --warmup 2 --skip-cpu --streams 1 > $OUT/pmc_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -- python3 bench.py --frames 16 --steps 8 --warmup 2 --skip-cpu --streams 1 > $OUT/pmc_lds.log 2>&1
find $OUT -name "*.csv" | head -20
