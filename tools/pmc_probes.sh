#!/bin/bash
# Instruction counts of the splat kernel's cumulative-stage probes (diagnostic library): bash tools/pmc_probes.sh "<counters>"
# probes: 11 load + filter, 12 + sort, 13 + sweep, 14 + merge (no compositor), 0 everything
set -e
CTR=${1:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 11 12 13 14 0; do
  OUT=gpurun_out/pmc_probe_$v
  rm -rf $OUT; mkdir -p $OUT
  EGOPRIOR_LIB=libegoprior_hip_tiletimes.so timeout -s KILL 120 rocprofv3 --pmc $CTR --output-format csv -d $OUT -- python3 bench.py --frames 16 --steps 8 --warmup 2 --skip-cpu --streams 1 --opt splat_variant=$v > $OUT/log.txt 2>&1
  echo "variant $v: $(python3 tools/pmc_summary.py $OUT | grep splat | sed 's/.*n= *[0-9]* //')"
done
