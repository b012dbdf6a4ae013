#!/bin/bash
# Collects the rocprofv3 evidence for one round, every pass on the SAME command and therefore the same frames as the
# timing: `python3 bench.py --steps 20 --warmup 5 --skip-cpu --streams 1` (the driver's steps / warmup; one frame in flight so
# that a launch's counters are its own).  Passes: kernel-trace stats, HBM read bytes, HBM write bytes, SQ instruction /
# wave / wait counters, LDS counters -- counters in runs of their own, never together with a trace.
# Run on the GPU box: bash tools/profile_round.sh r03 ; then here: python tools/make_profiles.py r03
set -e
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$R
rm -rf $OUT
mkdir -p $OUT
CMD="bench.py --steps 20 --warmup 5 --skip-cpu --streams 1"
echo "$CMD" > $OUT/command.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $CMD > $OUT/bench_trace.json 2> $OUT/bench_trace.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "ep::" --output-format csv -d $OUT/pmc_fetch -- python3 $CMD > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "ep::" --output-format csv -d $OUT/pmc_write -- python3 $CMD > $OUT/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-include-regex "ep::" --output-format csv -d $OUT/pmc_sq -- python3 $CMD > $OUT/pmc_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-include-regex "ep::" --output-format csv -d $OUT/pmc_lds -- python3 $CMD > $OUT/pmc_lds.log 2>&1
# the same command without a profiler, for the line the counters are compared with
timeout -k 10 200 python3 $CMD > $OUT/bench_plain.json 2> $OUT/bench_plain.err
# keep what goes back small: the per-kernel statistics of the trace, not the trace itself
find $OUT/trace -name "*kernel_trace.csv" -delete
find $OUT -name "*.csv" | head -20
du -sh $OUT
